set -o pipefail
mkdir -p gpurun_out/r03b
timeout -k 10 300 python -m pytest tests/test_batched.py -m gpu -x -q > gpurun_out/r03b/test_batched.log 2>&1; echo "test_batched rc=$?" | tee gpurun_out/r03b/status.txt
tail -15 gpurun_out/r03b/test_batched.log
grep -q "rc=0" gpurun_out/r03b/status.txt && timeout -k 10 400 python scripts/archive/sweep_mfma16.py 32768 100 > gpurun_out/r03b/sweep_32768.jsonl 2> gpurun_out/r03b/sweep.err; echo "sweep rc=$?" | tee -a gpurun_out/r03b/status.txt
cat gpurun_out/r03b/sweep_32768.jsonl
