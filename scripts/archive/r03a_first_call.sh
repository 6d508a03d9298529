set -o pipefail
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a gpurun_out/r03a/status.txt
tail -5 gpurun_out/r03a/gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a/bench20.json 2> gpurun_out/r03a/bench20.err; echo "bench rc=$?" | tee -a gpurun_out/r03a/status.txt
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 scripts/probes/rccl_two_ranks_one_gpu.py > gpurun_out/r03a/rccl2.log 2>&1; echo "rccl2 rc=$?" | tee -a gpurun_out/r03a/status.txt
tail -5 gpurun_out/r03a/rccl2.log
