N=${1:-20}
for i in $(seq 1 $N); do python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29700 + i)) tools/loud_detect.py 2>/dev/null | grep LOUD; done | grep -c BAD
