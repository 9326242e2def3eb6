#!/bin/bash
for f in 4 8 16 4 8 16; do
  echo "NC_S3X_FLUSH=$f"; NC_S3X_FLUSH=$f timeout 300 python3 tools/h2_check.py 2>&1 | grep "64->64 40^3 k3 relu\|108^3 k3\|108^3 k5" | sed 's/fp32mfma.*| fp16x2/fp16x2/' | cut -c1-200
done
