mkdir -p gpurun_out/r04u
python tools/sparse_share.py > gpurun_out/r04u/sparse_share.txt 2>&1
python bench.py > gpurun_out/r04u/bench_default.json 2> gpurun_out/r04u/bench_default.err
python bench.py --size 512 --steps 10 --no-secondary > gpurun_out/r04u/bench_512.json 2>/dev/null
python bench.py --data depth --no-secondary --no-cpu-baseline > gpurun_out/r04u/bench_depth_256.json 2>/dev/null
python bench.py --data depth --size 512 --steps 10 --no-secondary --no-cpu-baseline > gpurun_out/r04u/bench_depth_512.json 2>/dev/null
python bench.py --workload sobolev --no-cpu-baseline --steps 10 > gpurun_out/r04u/bench_sobolev.json 2>/dev/null
python tools/host_timeline.py 256 > gpurun_out/r04u/host_timeline_256.txt 2>&1
for args in "--size 64 --iterations 6 --halo 2" "--size 64 --iterations 6 --scaling strong" "--size 64 --iterations 6 --data depth"; do python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-device --no-cpu-baseline $args >> gpurun_out/r04u/bench_two_ranks_gloo.jsonl 2>/dev/null; done
cat gpurun_out/r04u/sparse_share.txt
