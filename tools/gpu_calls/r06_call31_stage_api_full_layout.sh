RT_FUZZ_SEEDS=5653 timeout 250 python -m pytest "tests/test_gpu_fuzz.py::test_random_scene_matches_oracle_bit_for_bit[5652]" -q -m gpu -p no:cacheprovider 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_frame_kernel.py tests/test_gpu_samples_ahead.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
