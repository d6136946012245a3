"""Experiment helper (round 4): patch in-kernel cycle stamps into csrc/wgrad_x9.hip (print once from the 40th eager launch); restore with git checkout."""
import sys
p='pytorch-ardae-vae_amd/csrc/wgrad_wide_tiles.h'
s=open(p).read()
s=s.replace('  int ntiles, splits;','  int ntiles, splits;\n  unsigned long long* stamps;')
open(p,'w').write(s)
p='pytorch-ardae-vae_amd/csrc/wgrad_x9.hip'
s=open(p).read()
s=s.replace('''      unsigned char* cur = lds + buf * X_BUF_BYTES;''','''      const bool stamp = batch.stamps && blockIdx.x == 7 && ch == c_begin + 20 && tid == 0;
      unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0, TM[9] = {0};
      if (batch.stamps) T0 = __builtin_readcyclecounter();
      unsigned char* cur = lds + buf * X_BUF_BYTES;''',1)
s=s.replace('''        constexpr int n = decltype(nn)::value, s = n >> 4, idx = n & 15, a = idx >> 2, b = idx & 3;''','''        constexpr int n = decltype(nn)::value, s = n >> 4, idx = n & 15, a = idx >> 2, b = idx & 3;
        if constexpr (n == 1) { if (batch.stamps) T1 = __builtin_readcyclecounter(); }
        if constexpr (idx == 15) { if (batch.stamps) TM[s] = __builtin_readcyclecounter(); }
        if constexpr (n == 143) { if (batch.stamps) T2 = __builtin_readcyclecounter(); }''',1)
s=s.replace('''      bsum += cs * c1.fb; rsum += cr * c1.fr;
      buf ^= 1;''','''      bsum += cs * c1.fb; rsum += cr * c1.fr;
      buf ^= 1;
      if (batch.stamps) T3 = __builtin_readcyclecounter();
      if (stamp) { batch.stamps[0] = T0; batch.stamps[1] = T1; batch.stamps[2] = T2; batch.stamps[3] = T3; batch.stamps[4] = __builtin_readcyclecounter(); batch.stamps[5] = __builtin_amdgcn_s_memrealtime();
                   for (int q = 0; q < 9; ++q) batch.stamps[8 + q] = TM[q]; }
      if (batch.stamps && blockIdx.x == 7 && ch == c_begin + 120 && tid == 0) { batch.stamps[6] = __builtin_readcyclecounter(); batch.stamps[7] = __builtin_amdgcn_s_memrealtime(); }''',1)
s=s.replace('''int launch_wgrad_x9(const WwBatchDev& b, hipStream_t st) {''','''int launch_wgrad_x9(const WwBatchDev& b0, hipStream_t st) {
  static unsigned long long* dstamps = nullptr;
  static int calls = 0;
  WwBatchDev b = b0;
  b.stamps = nullptr;
  if (debug_knob("ARDAE_X9_STAMPS")) {
    if (!dstamps) { (void)hipMalloc(&dstamps, 256); (void)hipMemset(dstamps, 0, 256); }
    b.stamps = dstamps;
    if (++calls == 40) {
      (void)hipDeviceSynchronize();
      unsigned long long h[32];
      (void)hipMemcpy(h, dstamps, 256, hipMemcpyDeviceToHost);
      fprintf(stderr, "x9 stamps (cycles): start->MFMA1 %llu, MFMA1->MFMA143 %llu, ->end of trip %llu, barrier %llu; 100 trips: %llu cycles, %llu ticks of 100 MHz; products:", h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[6] - h[4], h[7] - h[5]);
      for (int q = 0; q < 9; ++q) fprintf(stderr, " %llu", h[8 + q] - (q ? h[8 + q - 1] : h[1]));
      fprintf(stderr, "\\n");
    }
  }''',1)
open(p,'w').write(s)
