set -o pipefail
mkdir -p gpurun_out/r03n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03n/gputests.log 2>&1; echo "gpu tests rc=$?" | tee gpurun_out/r03n/status.txt
tail -3 gpurun_out/r03n/gputests.log
python bench.py > gpurun_out/r03n/bench_default.json 2> gpurun_out/r03n/bench_default.err; echo "bench default rc=$?" | tee -a gpurun_out/r03n/status.txt
bash scripts/archive/profile_bench.sh > gpurun_out/r03n/profile_bench.log 2>&1; echo "profile_bench rc=$?" | tee -a gpurun_out/r03n/status.txt
bash scripts/archive/profile_kernels.sh > gpurun_out/r03n/profile_kernels.log 2>&1; echo "profile_kernels rc=$?" | tee -a gpurun_out/r03n/status.txt
tail -25 gpurun_out/r03n/profile_kernels.log
