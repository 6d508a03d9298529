set -o pipefail
mkdir -p gpurun_out/r03l
timeout -k 10 600 python -m pytest tests/test_rls.py tests/test_abi.py -x -q -m gpu > gpurun_out/r03l/rls_tests.log 2>&1 || { tail -40 gpurun_out/r03l/rls_tests.log; exit 1; }
tail -2 gpurun_out/r03l/rls_tests.log
timeout -k 10 500 python scripts/archive/probes/rls_sym_cap.py > gpurun_out/r03l/cap.jsonl 2> gpurun_out/r03l/cap.err || { tail -20 gpurun_out/r03l/cap.err; exit 1; }
tail -2 gpurun_out/r03l/cap.jsonl
