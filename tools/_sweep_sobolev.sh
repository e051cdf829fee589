mkdir -p gpurun_out/r04i
for s in 8 16 32; do for b in 32 64 128 256; do
  echo -n "strips $s blocks_per_xcd $b: " >> gpurun_out/r04i/sobolev_sweep.txt
  LSF_SOBOLEV_STRIPS=$s LSF_LIST_BLOCKS_PER_XCD=$b python bench.py --workload sobolev --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r04i/sobolev_sweep.txt
done; done
