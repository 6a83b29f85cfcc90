for v in single overlap single overlap single overlap; do
  echo -n "allreduce=$v  "
  python bench.py --no-cpu-baseline --steps 60 --allreduce $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
echo "torchrun nproc 1:"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --allreduce overlap 2>&1 | tail -2 | cut -c1-400
