for mb in 1 2 4 8 16 32; do echo "slice $mb MiB"; SSW_COPY_SLICE_MB=$mb python tools/handle_bench.py 2>&1 | grep -E "4 copy|u8 pinned"; done
