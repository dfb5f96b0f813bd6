# writes phase clocks into jpeg_par_decode_kernel (a throw-away build: restore k_jpeg.hip afterwards)
import sys
p = 'vision_slam_frontend_amd/csrc/k_jpeg.hip'
s = open(p).read()
def sub(old, new):
    global s
    assert s.count(old) == 1, old[:50]
    s = s.replace(old, new)
sub('  // ---- 1. remove the byte stuffing: raw -> clean', '  long long T0 = wall_clock64(), T1, T2, T3, T4; int ROUNDS = 0;\n  // ---- 1. remove the byte stuffing: raw -> clean')
sub('  // ---- 2. segment end states until they stop changing ----', '  T1 = wall_clock64();\n  // ---- 2. segment end states until they stop changing ----')
sub('  for (int round = 0; round < kParThreads; round++) {\n    const uint32_t nq', '  T2 = wall_clock64();\n  for (int round = 0; round < kParThreads; round++) { ROUNDS++;\n    const uint32_t nq')
sub('  // ---- 3. block and DC offsets, then the writing pass ----', '  T3 = wall_clock64();\n  // ---- 3. block and DC offsets, then the writing pass ----')
sub('    par_write(G, W, tab, zz, s_blk + wid * 64 * kBlkStride, lane, limit, q, c, k, g, pred, coef32, total_blocks,\n              t == kParThreads - 1);\n  }\n}', '''    par_write(G, W, tab, zz, s_blk + wid * 64 * kBlkStride, lane, limit, q, c, k, g, pred, coef32, total_blocks,\n              t == kParThreads - 1);
  }
  __syncthreads();
  T4 = wall_clock64();
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) printf("blk %d len %u: destuff %lld first %lld rounds(%d) %lld write %lld (100MHz ticks)\\n", blockIdx.x, len, T1-T0, T2-T1, ROUNDS, T3-T2, T4-T3);
}''')
open(p, 'w').write(s)
