timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "batch_norm or gan_loss or apollo_step or partly_frozen" 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | cut -c1-300 | head -30
