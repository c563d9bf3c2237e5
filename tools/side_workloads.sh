for w in cornell glass4k emissive; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-primary > gpurun_out/side_$w.json 2>gpurun_out/side_$w.err || exit 1
done
timeout -k 10 200 python bench.py --pipeline mega --steps 1 --warmup 1 --spp 64 --no-cpu-baseline --no-primary > gpurun_out/side_mega.json 2>gpurun_out/side_mega.err
