python tools/probe/dbg5.py 2>&1 | grep -v amdgpu.ids | grep maxdiff
timeout 1500 python -m pytest tests/test_gpu_c8x.py tests/test_gpu_config3.py tests/test_gpu_fullsize.py::test_diced_inference_slab_mode_single_rank_is_bit_identical "tests/test_gpu_nets.py::test_apollo_step" "tests/test_gpu_nets.py::test_discriminators_wide" tests/test_gpu_nets.py::test_gan_loss_modes -q -m gpu 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | cut -c1-300 | head -30
timeout 300 python tools/c8x_time.py 5 2>&1 | grep "x "
