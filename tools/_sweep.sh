for g in "0 0" "8 0" "16 0" "2 0" "0 8" "0 2" "0 16"; do
  set -- $g
  echo "GM rows=$1 cols=$2"; SSW_GM_ROWS=$1 SSW_GM_COLS=$2 python tools/dct_microbench.py 3840 2160 64 2 f64 0
done
