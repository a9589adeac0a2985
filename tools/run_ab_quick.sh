# quick GPU check: parity suites, then the same-box bench alternation (tools/ab_fused.sh [variant libs])
mkdir -p gpurun_out/r2w
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py tests/test_gpu_train.py -q -x > gpurun_out/r2w/pytest.log 2>&1; tail -5 gpurun_out/r2w/pytest.log
bash tools/ab_fused.sh "$@" > gpurun_out/r2w/ab.log 2>&1; cat gpurun_out/r2w/ab.log
