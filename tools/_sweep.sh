for i in 1 2; do
python tools/handle_bench.py 2>&1 | grep -E "u8 pinned|^1 host"
SSW_NO_SPLIT=1 python tools/handle_bench.py 2>&1 | grep -E "u8 pinned|^1 host" | sed 's/^/NOSPLIT /'
done
