L="shipped tools/bin/libbt709hip_stag1.so tools/bin/libbt709hip_stag19.so tools/bin/libbt709hip_stag135.so"
python tools/ab_libs.py --contiguous --rounds 3 $L
python tools/ab_libs.py --contiguous --rounds 3 --decoder-option 5=0 --per-launch 32 shipped
for pad in 4096 65536 1048576 2097152 8192000; do python tools/ab_libs.py --contiguous --rounds 3 --out-pad $pad shipped; done
for pad in 4096 65536 1048576; do python tools/ab_libs.py --contiguous --rounds 3 --in-pad $pad shipped; done
for sh in 4096 65536 1048576; do python tools/ab_libs.py --contiguous --rounds 3 --out-shift $sh shipped; done
