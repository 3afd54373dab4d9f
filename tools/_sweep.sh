python tools/dct_microbench.py 3840 2160 64 2 f64 0
SSW_ONEBLOCK=1 python tools/dct_microbench.py 3840 2160 64 2 f64 0
