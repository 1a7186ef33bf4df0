export PD_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for e in 0 1; do
  s=$(date +%s)
  PD_WGRAD_SIDE=$e timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+e)) bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sweep --workload sd_train --batch 2 --size 16 2>/tmp/err_$e.txt | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); dp=j.get('data_parallel',{})
        print('value',j['value'],'ms_per_step',j['ms_per_step'],{k:dp[k] for k in dp if k!='per_rank'})
"
  echo "PD_WGRAD_SIDE=$e wall $(( $(date +%s) - s )) s"; tail -3 /tmp/err_$e.txt | cut -c1-200
done
