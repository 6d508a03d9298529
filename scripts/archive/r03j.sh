set -o pipefail
mkdir -p gpurun_out/r03j
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03j/gputests.log 2>&1; echo "gpu tests rc=$?" | tee gpurun_out/r03j/status.txt
tail -4 gpurun_out/r03j/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03j/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r03j/status.txt; tail -1 gpurun_out/r03j/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03j/bench_driver_cmd.json 2> gpurun_out/r03j/bench_driver_cmd.err; echo "bench20 rc=$?" | tee -a gpurun_out/r03j/status.txt
python bench.py > gpurun_out/r03j/bench_default.json 2> gpurun_out/r03j/bench_default.err; echo "bench default rc=$?" | tee -a gpurun_out/r03j/status.txt
wc -l gpurun_out/r03j/bench_driver_cmd.json gpurun_out/r03j/bench_default.json
