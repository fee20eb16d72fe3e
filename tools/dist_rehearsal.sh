python bench.py --steps 100 --warmup 20 --no-cpu-baseline >/dev/null 2>&1
for r in 0; do
for d in 0 1; do
  if [ $d = 1 ]; then export CTF_FORCE_DIST=1; else unset CTF_FORCE_DIST; fi
  CTF_OBS_RESERVE_BLOCKS=$r python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('reserve=$r dist=$d', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()})"
done; done
