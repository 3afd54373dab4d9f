"""Pins the CPU oracle (oracle/) against the reference's own known-answer tests.

Every case below is one of the reference's unit / integration tests with its
vectors transcribed in tests/golden/known_answers.json (file:line inside).  No
GPU and nothing from /root/reference is needed at run time.
"""
import numpy as np
import pytest

from oracle import oracle as O
from conftest import f32_to_u8, u8_to_f32

BACKENDS = [O.BACKEND_F64, O.BACKEND_F32, O.BACKEND_NAIVE_F64]


def approx_equal(a, b, tol):
    """util.rs:24-43: max abs error."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol, (a, b)


@pytest.mark.parametrize("backend", BACKENDS)
def test_simple_dct_against_scipy(known_answers, backend):
    ka = known_answers["dct1d_simple"]
    v = O.dct1d(ka["input"], O.DCT2, backend)
    approx_equal(v, ka["dct2_rustdct"], ka["tol"])
    back = O.dct1d(v, O.DCT3, backend) * np.float32(2.0 / 3.0)
    approx_equal(back, ka["input"], ka["tol"])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", ["dct2d_almost_identity", "dct2d_no_ones", "dct2d_larger"])
def test_2d_dct_against_scipy(known_answers, backend, case):
    ka = known_answers[case]
    x = np.array(ka["input"], np.float32).reshape(ka["h"], ka["w"])
    c = O.dct2d(x, O.DCT2, backend)
    approx_equal(c, ka["dct2"], ka["tol"])
    approx_equal(O.dct2d(c, O.DCT3, backend), x, ka["tol"])
    if "dct2_orthogonal" in ka:
        approx_equal(O.dct2d(x, O.DCT2_ORTHOGONAL, backend), ka["dct2_orthogonal"], ka["tol"])


@pytest.mark.parametrize("backend", BACKENDS)
def test_ortho_dct(known_answers, backend):
    ka = known_answers["dct1d_ortho_simple"]
    # a 1xN plane run through the 2-D ortho transform: the H=1 pass multiplies by sqrt(1/4)*2 = 1
    got = O.dct2d(np.array([ka["input"]], np.float32), O.DCT2_ORTHOGONAL, backend)
    approx_equal(got, ka["expected"], ka["tol"])
    ka = known_answers["dct2d_ortho_4x3"]
    x = np.array(ka["input"], np.float32).reshape(ka["h"], ka["w"])
    approx_equal(O.dct2d(x, O.DCT2_ORTHOGONAL, backend), ka["dct2_orthogonal"], ka["tol"])


def test_backends_agree_on_odd_shapes():
    rng = np.random.default_rng(5)
    for h, w in [(1, 7), (7, 1), (16, 24), (37, 74), (135, 240), (111, 160)]:
        x = rng.random((h, w)).astype(np.float32)
        ref = O.dct2d(x, O.DCT2, O.BACKEND_NAIVE_F64)
        fft64 = O.dct2d(x, O.DCT2, O.BACKEND_F64)
        fft32 = O.dct2d(x, O.DCT2, O.BACKEND_F32)
        scale = np.abs(ref).max()
        assert np.abs(fft64 - ref).max() <= 2e-7 * scale
        assert np.mean(fft64 == ref) > 0.98          # both are the correctly rounded transform
        assert np.abs(fft32 - ref).max() <= 2e-6 * scale
        back = O.dct2d(fft64, O.DCT3, O.BACKEND_F64)
        assert np.abs(back - x).max() < 1e-6


def test_dct_matches_scipy_when_available():
    sf = pytest.importorskip("scipy.fft")
    rng = np.random.default_rng(6)
    x = rng.random((60, 100)).astype(np.float32)
    ref = sf.dct(sf.dct(x.astype(np.float64), axis=1), axis=0)
    got = O.dct2d(x, O.DCT2, O.BACKEND_F64)
    assert np.abs(got - ref).max() <= 2e-7 * np.abs(ref).max()
    # unnormalised DCT-III on both axes / (4 W H) inverts
    inv = sf.dct(sf.dct(ref, type=3, axis=1), type=3, axis=0) / (4 * 60 * 100)
    assert np.abs(O.dct2d(got, O.DCT3, O.BACKEND_F64) - inv).max() < 1e-6


def test_yiq_to_rgb(known_answers):
    ka = known_answers["yiq_triples"]
    for p in ka["pairs"]:
        rgb = np.array(p["rgb"], np.float32).reshape(1, 1, 3)
        y, i, q = O.rgb_to_yiq(rgb)
        approx_equal([y[0, 0], i[0, 0], q[0, 0]], p["yiq"], ka["tol"])
        yiq = np.array(p["yiq"], np.float32)
        back = O.yiq_to_rgb(yiq[0].reshape(1, 1), yiq[1].reshape(1, 1), yiq[2].reshape(1, 1))
        approx_equal(back, p["rgb"], ka["tol"])


def test_yiq_to_rgb_image(known_answers):
    ka = known_answers["yiq_image_5x5"]
    img = np.zeros((ka["h"], ka["w"], 3), np.float32)
    for x, y, rgb in ka["pixels_xy_rgb"]:
        img[y, x] = rgb
    back = O.yiq_to_rgb(*O.rgb_to_yiq(img))
    approx_equal(back, img, ka["tol"])


def test_yiq_clamps():
    rgb = O.yiq_to_rgb(np.array([[1.0, 0.0]], np.float32), np.array([[0.5, -0.5]], np.float32),
                       np.array([[0.5, 0.5]], np.float32))
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0
    assert rgb[0, 0, 0] == 1.0 and rgb[0, 1, 1] == 0.0


def test_indices(known_answers):
    ka = known_answers["indices"]
    c = np.array(ka["coefficients"], np.float32)
    assert O.indices(c).tolist() == ka["expected"]
    for k in range(1, 6):                                   # partial path == prefix of the full sort
        assert O.indices(c, k=k).tolist() == ka["expected"][:k]


def test_indices_ties_are_stable_and_partial_equals_full():
    rng = np.random.default_rng(7)
    c = rng.integers(-6, 7, size=(12, 17)).astype(np.float32)      # many exact ties, +/- pairs
    for ordering in (O.ORDER_ENERGY, O.ORDER_ENERGY_ORTHOGONAL, O.ORDER_LEGACY):
        full = O.indices(c, ordering)
        assert sorted(full.tolist()) == list(range(1, c.size))
        keys = O.order_keys(c, ordering)
        kf = keys[full.astype(np.int64)]
        assert np.all(kf[:-1] >= kf[1:])
        same = kf[:-1] == kf[1:]
        assert np.all(full[:-1][same] < full[1:][same])             # ties: lower index first
        for k in (1, 5, 50, 150, c.size - 2):
            assert np.array_equal(O.indices(c, ordering, k=k), full[:k])


def test_insert_extract_functions(known_answers):
    ka = known_answers["insert_extract"]
    c = np.array(ka["coefficients"], np.float32)
    m = np.array(ka["mark"], np.float32)
    idx = np.arange(6, dtype=np.uint64)
    for method in (O.OPTION1, O.OPTION2, O.OPTION3):
        emb = O.embed(c, idx, [m], method, ka["alpha"])
        # k must be < n (algorithm.rs:553-555): append a dummy coefficient
        ext = O.extract(np.append(c, 1), np.append(emb, 1), idx, 6, method, ka["alpha"])
        approx_equal(ext, m, ka["tol"])


def _expected_single(a=np.float32(0.1)):
    f = np.float32
    return np.array([f(-3), f(5) * (f(1) + f(1) * a), f(-8) * (f(1) + f(1) * a),
                     f(7) * (f(1) - f(0.5) * a), f(1), f(2)], np.float32)


def test_embedder_single(known_answers):
    ka = known_answers["embedder_single"]
    c = np.array(ka["coefficients"], np.float32)
    idx = O.indices(c)
    emb = O.embed(c, idx, [np.array(ka["mark"], np.float32)], O.OPTION2, ka["alpha"])
    assert np.array_equal(emb, _expected_single())           # assert_eq!: exact
    ext = O.extract(c, emb, idx, 3, O.OPTION2, ka["alpha"])
    assert np.abs(ext - np.array(ka["mark"], np.float32)).max() < ka["extract_tol"]


def test_embedder_single_and_zero(known_answers):
    ka = known_answers["embedder_single_and_zero"]
    c = np.array(ka["coefficients"], np.float32)
    emb = O.embed(c, O.indices(c), [np.array(m, np.float32) for m in ka["marks"]], O.OPTION2, ka["alpha"])
    assert np.array_equal(emb, _expected_single())


def test_embedder_multiple(known_answers):
    ka = known_answers["embedder_multiple"]
    c = np.array(ka["coefficients"], np.float32)
    emb = O.embed(c, O.indices(c), [np.array(m, np.float32) for m in ka["marks"]], O.OPTION2, ka["alpha"])
    f, a = np.float32, np.float32(0.1)
    def upd(x, w1, w2):
        d1 = f(x) * (f(1) + f(w1) * a) - f(x)
        d2 = f(x) * (f(1) + f(w2) * a) - f(x)
        return f(x) + d1 + d2
    expected = np.array([f(-3), upd(5, 1.0, -1.0), upd(-8, 1.0, 0.5), upd(7, -0.5, -0.5), f(1), f(2)], np.float32)
    assert np.array_equal(emb, expected)


def test_extract_error_behaviour():
    c = np.arange(6, dtype=np.float32)
    with pytest.raises(ValueError, match="length not equal"):
        O.extract(c, c[:5], np.arange(1, 4), 3)
    with pytest.raises(ValueError, match="exceeds available"):
        O.extract(c, c, np.arange(1, 6), 6)


def test_mark_longer_than_coefficients_is_truncated():
    c = np.array([-3, 5, -8, 7], np.float32)
    emb = O.embed(c, O.indices(c), [np.ones(10, np.float32)], O.OPTION1, 0.5)   # zip() truncation
    assert np.array_equal(emb, np.array([-3, 5.5, -7.5, 7.5], np.float32))


def test_similarity_sequential_f32():
    rng = np.random.default_rng(3)
    e, m = rng.standard_normal(1000).astype(np.float32), rng.standard_normal(1000).astype(np.float32)
    nom = den = np.float32(0)
    for a, b in zip(e, m):
        nom = np.float32(nom + np.float32(a * b))
        den = np.float32(den + np.float32(a * a))
    assert O.similarity(e, m) == np.float32(nom / np.sqrt(den))


def test_fixed_marks_are_standard_normal(marks):
    m = marks["seed_1"]
    assert m.shape == (1000,) and abs(float(m.mean())) < 0.1 and abs(float(m.std()) - 1) < 0.05
    assert np.allclose(m[:4], [-0.23484705, -1.4108177, 0.33302864, -1.1267663], atol=1e-7)


def test_single_simple_flow(known_answers, marks, cat_images):
    """tests/single_simple.rs on the oracle, self-consistent decode."""
    th = known_answers["single_simple_thresholds"]
    cat = u8_to_f32(cat_images["cat"])
    mark = marks["seed_1"]
    wm8 = f32_to_u8(O.embed_frame(cat, mark))
    ext, sim = O.extract_frame(cat, u8_to_f32(wm8), mark)
    assert np.abs(ext - mark).max() < 0.16          # reference: 0.12 with its own JPEG decoder
    assert np.abs(ext - mark).mean() < th["mean_err"]
    assert sim > th["sim_gt"]
    assert O.similarity(ext, marks["seed_baaaaaad"]) < th["random_sim_lt"]


def test_seed1_mark_is_the_one_in_the_reference_png(marks, cat_images):
    """The reference's fixture watermarked_with_1.png carries generate_fixed_normal_sequence(1, 1000):
    extracting from it with the oracle must correlate with our restated ChaCha8/ziggurat mark and
    with no other.  (Not pixel-exact reproducible: PIL's JPEG decode differs from Rust's.)"""
    ext, sim = O.extract_frame(u8_to_f32(cat_images["cat"]), u8_to_f32(cat_images["watermarked_with_1"]),
                               marks["seed_1"])
    assert sim > 15.0
    assert abs(O.similarity(ext, marks["seed_2"])) < 3.0
    assert abs(O.similarity(ext, marks["seed_baaaaaad"])) < 3.0
    assert 30.0 < np.linalg.norm(ext) < 34.0        # sqrt(X*.X*) of an N(0,1) mark of length 1000
