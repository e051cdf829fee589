mkdir -p gpurun_out/r04j
run() { name=$1; shift; python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-device --no-cpu-baseline "$@" > gpurun_out/r04j/$name.out 2> gpurun_out/r04j/$name.err; echo "$name rc=$?" >> gpurun_out/r04j/summary.txt; }
run weak_faces --size 64 --iterations 6 --halo 2
run weak_centered --size 96 --iterations 4 --halo 2 --pattern centered
run strong --size 64 --iterations 6 --scaling strong
run depth --size 64 --iterations 6 --data depth
run depth128 --size 128 --iterations 6 --data depth
