for g in 512 256 128; do echo "== grid $g"; NC_C8X_GRID=$g timeout 120 python tools/c8x_time.py 5 2 2 2>&1 | grep "x "; done
