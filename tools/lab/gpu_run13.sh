#!/bin/bash
# round 2, pass 13: device base selection: new parity tests, the configs that use it by size, timing
mkdir -p gpurun_out
echo "== select tests"
timeout 900 python -m pytest tests/test_gpu_select.py -x -q 2>&1 | tail -15
echo "== config4 + registration + sharding tests (device selection by size / selector thread)"
timeout 1500 python -m pytest tests/test_gpu_configs.py::test_config4_part_in_whole_10m_scene tests/test_gpu_registration.py tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | tail -8
echo "== timing"
timeout 900 python tools/f3_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_f3_timing.log
