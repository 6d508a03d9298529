set -o pipefail
mkdir -p gpurun_out/r03d
for N in 4096 8192 16384 24576 65536; do
  SIZES=1 timeout -k 10 300 python scripts/archive/sweep_mfma16.py $N 100 > gpurun_out/r03d/sizes_$N.jsonl 2>> gpurun_out/r03d/err.log || echo "N=$N failed"
  echo "== N=$N"; cut -c1-140 gpurun_out/r03d/sizes_$N.jsonl
done
