set -e
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/sdprof
for dt in bf16 fp16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sd_$dt -- python3 bench.py --workload sd_img2img --dtype $dt --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-side-workloads > gpurun_out/sdprof/bench_$dt.log 2>gpurun_out/sdprof/bench_$dt.err
  cp "$(ls /tmp/prof_sd_$dt/*/*_kernel_stats.csv | head -1)" gpurun_out/sdprof/sd_img2img_${dt}_kernel_stats.csv
  tail -c 400 gpurun_out/sdprof/bench_$dt.log | head -c 400; echo
done
