set -o pipefail
mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03e/gputests.log 2>&1; echo "gpu tests rc=$?" | tee gpurun_out/r03e/status.txt
tail -4 gpurun_out/r03e/gputests.log
bash scripts/archive/profile_batched.sh > gpurun_out/r03e/profile_batched.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r03e/status.txt
tail -6 gpurun_out/r03e/profile_batched.log | cut -c1-400
