#!/bin/bash
# CPU sanitizer pass (GPU sanitizers are not available on this pool).  Runs in the build container; no GPU needed.
#   1. the oracle and the product's HOST-side table builders under -fsanitize=address,undefined
#   2. (round 5) the host shim itself -- csrc/shim_*.cpp + bt709_ring.cpp: coalescing queues, pools, sharder, ring hunts, ring
#      sets -- compiled against the tests-only fake HIP runtime (tests/native/fake_hip/) and driven by tests/native/shim_stress.cpp
#      under ASan + UBSan + LeakSanitizer and under TSan (the same two builds tests/test_fake_hip.py runs)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
gcc -std=gnu99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared \
    oracle/bt709_oracle.c -o $T/liboracle_asan.so -lm -lpthread
cat > $T/tables.cpp <<'CPP'
#include "metalbt709decoder_amd/csrc/transfer_tables.h"
#include <cstdio>
using namespace bt709;
int main() {
  for (int g = 0; g < kTableKinds; ++g) {
    TransferTable t; SplitTable s; UniformTable u;
    if (!build_transfer_table(g, &t) || !build_split_table(g, &s) || !build_uniform_table(g, 256, &u)) return 1;
    for (uint32_t bits = 0; bits <= 0x3f800000u; bits += 65537) {
      float x; __builtin_memcpy(&x, &bits, 4);
      const TransferBucket &b = t.buckets_unit[bucket_index(x, 8388608.0f / t.n)];
      const TransferBucket &c = u.buckets[uniform_index(x, (float)u.n)];
      if ((int)(b.base + (x >= b.edge)) != transfer_to_byte(g, x) || (int)(c.base + (x >= c.edge)) != transfer_to_byte(g, x)) return 2;
    }
  }
  for (int g = 0; g < kGammaCount; ++g) { HalfTable h; if (!build_half_table(g, &h)) return 3; }
  EncodeTables e; if (!build_encode_tables(kGammaSRGB, kGammaApple, &e)) return 4;
  std::puts("host tables ok under asan+ubsan");
  return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I. $T/tables.cpp \
    metalbt709decoder_amd/csrc/transfer_tables.cpp -o $T/tables && $T/tables
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 BT709_ORACLE_SO=$T/liboracle_asan.so \
    python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, "tests")
import numpy as np
import oracle_lib
oracle_lib.ORACLE_SO = os.environ["BT709_ORACLE_SO"]
oracle_lib.build_oracle = lambda force=False: None
o = oracle_lib.Oracle()
rng = np.random.default_rng(1)
for w, h in ((2, 2), (6, 4), (66, 12), (256, 16)):
    y = rng.integers(0, 256, (h, w), dtype=np.uint8); c = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for g in range(4):
        o.decode_nv12(g, y, c, alpha=a); o.decode_nv12_rgba16f(g, y, c, alpha=a)
        o.decode_nv12_scaled(g, y, c, 7, 5, alpha=a); o.render_scaled(o.decode_nv12(g, y, c), 9, 3)
        o.render_scaled(o.decode_nv12_rgba16f(g, y, c), 3, 9)
        if w % 4 == 0 and h % 4 == 0: o.decode_nv12_half(g, y, c, alpha=a)
print("oracle ok under asan+ubsan")
PY
SHIM="metalbt709decoder_amd/csrc/shim_core.cpp metalbt709decoder_amd/csrc/shim_decode.cpp metalbt709decoder_amd/csrc/shim_convert.cpp metalbt709decoder_amd/csrc/shim_coalesce.cpp metalbt709decoder_amd/csrc/shim_pool_shard.cpp metalbt709decoder_amd/csrc/shim_introspect.cpp metalbt709decoder_amd/csrc/bt709_ring.cpp metalbt709decoder_amd/csrc/transfer_tables.cpp tests/native/fake_hip/fake_hip.cpp tests/native/shim_stress.cpp"
CXXF="-std=c++17 -O1 -g -fno-omit-frame-pointer -Wall -Wno-format-truncation -Itests/native/fake_hip -Itests/native"
g++ $CXXF -fsanitize=address,undefined $SHIM -o $T/shim_stress_asan -lpthread
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $T/shim_stress_asan
g++ $CXXF -fsanitize=thread $SHIM -o $T/shim_stress_tsan -lpthread
for round in 1 2 3; do TSAN_OPTIONS=halt_on_error=1:exitcode=66 $T/shim_stress_tsan | tail -1; done
echo "host shim ok under asan+ubsan+lsan and tsan (fake HIP runtime)"
rm -rf $T
