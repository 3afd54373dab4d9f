python - <<PY
import ctypes as C, time
hip=C.CDLL("/opt/rocm/lib/libamdhip64.so")
ev=C.c_void_p(); hip.hipEventCreate(C.byref(ev))
t0=time.perf_counter()
for _ in range(2000): hip.hipEventRecord(ev, None); hip.hipEventQuery(ev)
print("hipEventRecord+Query: %.1f us" % ((time.perf_counter()-t0)/2000*1e6))
PY
for mb in 4 8 16; do echo "slice $mb MiB"; SSW_COPY_SLICE_MB=$mb python tools/handle_bench.py 2>&1 | grep -E "4 copy|u8 pinned"; done
