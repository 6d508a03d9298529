set -o pipefail
mkdir -p gpurun_out/r03f
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03f/bench_driver_cmd.json 2> gpurun_out/r03f/bench_driver_cmd.err; echo "bench20 rc=$?" | tee gpurun_out/r03f/status.txt
tail -3 gpurun_out/r03f/bench_driver_cmd.err
python bench.py > gpurun_out/r03f/bench_default.json 2> gpurun_out/r03f/bench_default.err; echo "bench default rc=$?" | tee -a gpurun_out/r03f/status.txt
tail -3 gpurun_out/r03f/bench_default.err
