#!/bin/bash
# N > 1 control flow of bench.py on ONE GPU: 2 and 4 ranks over gloo (PD_BENCH_REHEARSAL=1 -- RCCL refuses two ranks per
# device), all four workloads at small shapes, the driver's own command form.  Never a measurement: the lines say `rehearsal`.
# Fresh child processes only (torch.distributed.run starts the ranks); at most 4 ranks use the card at once.
cd "$(dirname "$0")/.."
OUT=gpurun_out/rehearsal
mkdir -p $OUT
export PD_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
port=29610
for n in 2 4; do
  for wl in img2img train sd_img2img sd_train; do
    case $wl in
      img2img) extra="--batch 4 --size 64 --inference-steps 3" ;;
      train) extra="--workload train --batch 8 --size 32" ;;
      sd_img2img) extra="--workload sd_img2img --batch 2 --size 128 --inference-steps 2" ;;
      sd_train) extra="--workload sd_train --batch 2 --size 16" ;;
    esac
    port=$((port + 1))
    timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 2 --warmup 1 \
      --no-cpu-baseline --no-roofline --no-sweep $extra > $OUT/r6_rehearsal_${wl}_n$n.json 2> $OUT/r6_rehearsal_${wl}_n$n.err
    rc=$?
    echo "$wl n=$n rc=$rc $(python -c "import json,sys; ls=[l for l in open('$OUT/r6_rehearsal_${wl}_n$n.json') if l.startswith('{')]; j=json.loads(ls[-1]); print(len(ls),'line(s): value',j['value'],j['unit'],'n_gpus',j['n_gpus'],'world',j.get('rccl_world_size'),'ranks',len(j.get('per_rank_units_per_s',[])), 'selftest exact', j.get('allreduce_selftest',{}).get('exact'))" 2>&1 | tail -1)"
    [ $rc = 0 ] || tail -5 $OUT/r6_rehearsal_${wl}_n$n.err
  done
done
