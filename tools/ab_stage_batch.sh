OLD="--library tools/bin/libbt709hip_before_batch.so"
for rep in 1 2; do
for lib in "$OLD" ""; do
  echo "== ${lib:-new}"
  python tools/bench_encode.py --frames-per-launch 1 $lib | cut -c1-210
  python tools/bench_encode.py --frames-per-launch 32 $lib | cut -c1-210
  python tools/bench_scaled.py --path render8 $lib | cut -c1-200
  python tools/bench_scaled.py --path render8 --frames-per-launch 16 $lib | cut -c1-200
  python tools/bench_scaled.py --path render16 --frames-per-launch 16 $lib | cut -c1-200
  python tools/bench_scaled.py --path scaled --frames-per-launch 1 $lib | cut -c1-200
  python tools/bench_scaled.py --path scaled --frames-per-launch 8 $lib | cut -c1-200
  python tools/bench_scaled.py --path scaled --width 1920 --height 1080 --out-width 3840 --out-height 2160 --ring 128 --frames-per-launch 8 $lib | cut -c1-200
done; done
