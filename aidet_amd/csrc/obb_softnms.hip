// obb_softnms.hip -- Soft-NMS on the GPU for gfx950.
//
// Replaces soft_nms_cpu_kernel (mmdet/ops/nms/src/nms_cpu.cpp:71-210), which the reference
// only offers on the CPU (nms_wrapper.py:64 "Dispatch to only CPU Soft NMS implementations").
//
// Soft-NMS is a chain of n DEPENDENT rounds: pick the best remaining box, decay the scores
// of the others by their overlap with it, drop what falls below min_score.  The reference
// keeps the boxes in one array it permutes in place, and that permutation is observable:
//   * the arg-max uses a STRICT `<` (:114-121), so among equal scores the box at the LOWEST
//     POSITION of the current array wins -- positions, not original indices;
//   * the pick is swapped with position i (:123-140);
//   * a dropped box is overwritten by the LAST box of the array, which is then examined at
//     that position (:184-198: pos is decremented and incremented again).
// One workgroup replays exactly that array:
//   round i:  (1) block-wide arg-max over positions [i, nd)  -> swap with i
//             (2) every position p > i: overlap with box i in the reference's own
//                 arithmetic (nms_cpu.cpp:155-163 = hbb_overlap), decay, drop flag
//             (3) the drops are compacted the way the sequential loop would have done it:
//                 with M survivors the new end is nd' = i + 1 + M; the holes (dropped
//                 positions below nd', ascending) are filled by the survivors at or above
//                 nd' taken from the END downwards -- the k-th hole receives the k-th
//                 survivor counted from the end.  (Sequentially: the loop reaches the holes
//                 in ascending order and each time pulls the current last element; a pulled
//                 element that is itself dropped is pulled over again, i.e. skipped.)
// A round without drops costs TWO barriers: the decay pass also votes on the next pick (valid when
// no position moves).  State (7 words per box) lives in LDS up to 2048 boxes and in the caller's
// workspace above that.
#include "obb_common.h"
#include "obb_geom.h"

namespace obb {

constexpr int kSoftLds = 2048;   // boxes whose state fits the LDS arrays below (56 KiB)
// A round is a chain of ~10 dependent LDS round trips and two block barriers around a few
// instructions per box: 1.2 - 2.0 us per round whatever the reductions are made of (shuffles ->
// DPP moves: -6 %; eight barriers -> two: -8 %), i.e. about the speed of the CPU loop it replaces
// (3.4 ms vs 3.3 ms at n = 2000 on this host).  The op exists so that `type='soft_nms'` works on
// device tensors without a round trip to the host, not for speed.

struct SoftState {
  float *x1, *y1, *x2, *y2, *sc, *ar;
  int *id;
};

// The pick of a round = the maximum of a 64-bit key over the remaining positions:
//   high word: order-preserving image of the score (-0 folded into +0; 0 for a NaN, which
//              never wins: nms_cpu.cpp:116 `max_score < scores[pos]` is false for it)
//   low word : ~position, so that among equal scores the LOWEST position wins (:114-121)
// A key of 0 means "nothing comparable".  Maxima of keys are reduced inside a wave with DPP
// moves (a shuffle is an LDS-crossbar round trip: twelve of them were ~1/3 of a round).
__device__ __forceinline__ unsigned long long pick_key(float s, int p) {
  const float t = s + 0.0f;
  uint32_t u = __float_as_uint(t);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  if (s != s) u = 0u;
  return ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)p);
}
__device__ __forceinline__ int key_pos(unsigned long long k) { return (int)(~(uint32_t)k); }
__device__ __forceinline__ bool key_none(unsigned long long k) { return (k >> 32) == 0ull; }

// The gaussian weight exp(-ovr^2 / sigma) (nms_cpu.cpp:176-178: std::exp on scalar_t = the host libm's expf for float
// rows).  The device library's expf is within one ulp of it -- and one ulp in a weight can turn two picks around or move a
// score across min_score (seen in the soak: another kept set).  Rounding the double exponential (correctly rounded) agrees
// with glibc in all but 0.06 % of the values -- still a few scores per array of thousands of decays.  So this is glibc's
// own algorithm restated (expf of glibc >= 2.27 = ARM optimized-routines: exp(x) = 2^(k/32) * 2^(r/32), the first from a
// table of 32 doubles, the second a cubic in double, one rounding to float at the end); tests/test_expf_restatement.py compares the
// restatement with the libm of this image bit for bit.  exp(-0) = 1 without the call: every box that misses the pick.
__device__ const unsigned long long kExpfTab[32] = {      // asuint64(2^(i/32)) - (i << 52) / 32
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
__device__ __forceinline__ float soft_exp(float x) {
  if (x == 0.0f) return 1.0f;
  if (x != x) return x + x;
  if (x < -0x1.9fe368p6f) return 0.0f;                      // (underflow; the callers' arguments are <= 0)
  if (x > 0x1.62e42ep6f) return __builtin_inff();
  const double inv_ln2_n = 0x1.71547652b82fep+0 * 32.0, shift = 0x1.8p+52;
  const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  const double z = inv_ln2_n * (double)x;
  double kd = z + shift;                                     // (k in the low bits of the sum: no contraction, no reassociation in this build)
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= shift;
  const double r = z - kd;
  const double sc = __longlong_as_double((long long)(kExpfTab[ki & 31ull] + (ki << 47)));
  const double q = c0 * r + c1;
  const double r2 = r * r;
  double y = c2 * r + 1.0;
  y = q * r2 + y;
  y = y * sc;
  return (float)y;
}
// float64 rows: std::exp on double = glibc's exp (the same family: 2^(k/128) as H[k] (1 + T[k]) from a table, a quintic in
// r), in the form the libm of an x86-64 host WITH fused multiply-add runs it (glibc selects its FMA build there: every
// a * b + c of the routine is one fma -- the uncontracted form differs in the last bit for 0.07 % of the arguments;
// tests/test_expf_restatement.py checks both statements against the libm of this image).  |x| >= 512 (sigma < 0.002):
// the device library's exp, within an ulp.
__device__ const unsigned long long kExpTab[256] = {       // asuint64(T[k]), asuint64(H[k]) - (k << 52) / 128
    0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
    0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
    0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
    0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
    0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
    0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
    0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
    0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
    0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
    0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
    0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
    0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
    0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
    0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
    0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
    0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
    0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
    0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
    0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
    0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
    0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
    0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
    0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
    0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
    0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
    0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
    0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
    0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
    0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
    0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
    0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
    0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
    0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
    0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
    0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
    0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
    0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
    0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
    0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
    0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
    0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
    0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
    0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
    0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
    0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
    0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
    0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
    0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
    0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
    0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
    0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
    0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
    0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
    0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
    0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
    0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
    0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
    0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
    0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
    0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
    0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
    0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
    0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
    0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull,
};
__device__ __forceinline__ double soft_exp(double x) {
  if (x == 0.0) return 1.0;
  if (x != x) return x + x;
  const double ax = x < 0.0 ? -x : x;
  if (ax < 0x1p-54) return 1.0 + x;
  if (ax >= 512.0) return exp(x);
  const double inv_ln2_n = 0x1.71547652b82fep0 * 128.0, shift = 0x1.8p52;
  const double neg_ln2_hi_n = -0x1.62e42fefa0000p-8, neg_ln2_lo_n = -0x1.cf79abc9e3b3ap-47;
  const double c2 = 0x1.ffffffffffdbdp-2, c3 = 0x1.555555555543cp-3, c4 = 0x1.55555cf172b91p-5, c5 = 0x1.1111167a4d017p-7;
  double kd = __builtin_fma(inv_ln2_n, x, shift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= shift;
  const double r = __builtin_fma(kd, neg_ln2_lo_n, __builtin_fma(kd, neg_ln2_hi_n, x));
  const unsigned long long idx = 2ull * (ki & 127ull);
  const double tail = __longlong_as_double((long long)kExpTab[idx]);
  const double scale = __longlong_as_double((long long)(kExpTab[idx + 1ull] + (ki << 45)));
  const double r2 = r * r;
  const double a = __builtin_fma(r, c3, c2), b = __builtin_fma(r, c5, c4);
  const double s1 = __builtin_fma(r2, a, tail + r);
  const double tmp = __builtin_fma(r2 * r2, b, s1);
  return __builtin_fma(scale, tmp, scale);
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max(unsigned long long v) {
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)v, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true);
  const unsigned long long o = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
  return o > v ? o : v;
}
// maximum over the 64 lanes, returned in every lane
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
  v = dpp_max<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
  v = dpp_max<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
  v = dpp_max<0x141>(v);   // row_half_mirror: 7 - lane inside a group of 8
  v = dpp_max<0x140>(v);   // row_mirror: 15 - lane inside a row of 16 -> every lane holds its row's maximum
  unsigned long long m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, r * 16);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), r * 16);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    m = o > m ? o : m;
  }
  return m;
}

// GROUPED calls (`seg` != nullptr): workgroup g replays the reference's loop on the rows [seg[g], seg[g + 1]) as an
// array of its own -- exactly what the per-class Python loop of multiclass_nms_with_index does with one soft_nms
// call per class (rbbox_nms.py:29-49) -- so the classes of a detection head run CONCURRENTLY (one workgroup
// each) instead of back to back with a host synchronisation per class.  Outputs of group g: rows
// [seg[g], seg[g] + result[g]) of out_dets / out_inds (indices relative to the group's first row).
template <bool LDS, int kSoftThreads>
__global__ __launch_bounds__(kSoftThreads) void soft_nms_kernel(const float *__restrict__ dets, int n,
                                                                float thr, int method, float sigma,
                                                                float min_score, float *__restrict__ ws,
                                                                float *__restrict__ out_dets,
                                                                int64_t *__restrict__ out_inds,
                                                                long long *__restrict__ result,
                                                                const int64_t *__restrict__ seg) {
  if (seg) {
    const bool only_given_up = n < 0;      // (n = -1: only the groups the position-free kernel gave up on, kSoftFastGiveUp in their count)
    const int64_t off = seg[blockIdx.x];
    n = (int)(seg[blockIdx.x + 1] - off);
    dets += off * 5;
    out_dets += off * 5;
    out_inds += off;
    ws += off * 8;
    result += blockIdx.x;
    if (only_given_up && result[0] != -2) return;                  // (block-uniform; kSoftFastGiveUp)
    if (n <= 0) { if (threadIdx.x == 0) result[0] = 0; return; }   // (block-uniform)
    if (LDS && n > kSoftLds) { if (threadIdx.x == 0) result[0] = -1; return; }   // (the host routes such calls to the workspace variant)
  }
  __shared__ float lds[LDS ? 7 * kSoftLds : 7];
  constexpr int kW = kSoftThreads / 64;
  __shared__ unsigned long long wbest[kW];
  __shared__ int wcnt[kW];
  __shared__ int holes[LDS ? kSoftLds : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cap = LDS ? kSoftLds : n;
  float *base = LDS ? lds : ws;
  SoftState S;
  S.x1 = base; S.y1 = base + cap; S.x2 = base + 2 * cap; S.y2 = base + 3 * cap;
  S.sc = base + 4 * cap; S.ar = base + 5 * cap; S.id = reinterpret_cast<int *>(base + 6 * cap);
  int *hole = LDS ? holes : reinterpret_cast<int *>(ws + 7 * (size_t)n);
  for (int p = tid; p < n; p += kSoftThreads) {
    const float *d = dets + (size_t)p * 5;
    const float a = d[0], b = d[1], c = d[2], e = d[3];
    S.x1[p] = a; S.y1[p] = b; S.x2[p] = c; S.y2[p] = e; S.sc[p] = d[4];
    S.ar[p] = hbb_area(a, b, c, e);
    S.id[p] = p;
  }
  __syncthreads();
  int nd = n;
  // block-wide arg-max over [from, nd): the first of equal scores, NaN never wins unless it sits
  // at `from` (the starting value of the reference's scan, nms_cpu.cpp:100-121); two barriers
  auto argmax_from = [&](int from) -> int {
    unsigned long long b = 0;
    for (int p = from + tid; p < nd; p += kSoftThreads) { const unsigned long long k = pick_key(S.sc[p], p); b = k > b ? k : b; }
    b = wave_max_key(b);
    __syncthreads();                 // (wbest may still be read from the previous use)
    if (lane == 0) wbest[wave] = b;
    __syncthreads();
    unsigned long long g = wbest[0];
#pragma unroll
    for (int k = 1; k < kW; ++k) g = wbest[k] > g ? wbest[k] : g;
    const float s0 = S.sc[from];
    return (s0 != s0 || key_none(g)) ? from : key_pos(g);
  };
  int mp = nd > 0 ? argmax_from(0) : 0;
  for (int i = 0; i < nd; ++i) {
    // ---- (1) the pick goes to position i
    __syncthreads();                 // every thread is done with the previous round's state
    if (tid < 7 && mp != i) {        // swap the 7 words of positions i and mp (bit patterns: one is an index)
      uint32_t *arr = reinterpret_cast<uint32_t *>(base) + (size_t)tid * cap;
      const uint32_t t = arr[i];
      arr[i] = arr[mp];
      arr[mp] = t;
    }
    __syncthreads();
    // ---- (2) decay of every later position; the SAME pass finds the next pick among the
    // survivors (valid when nothing is dropped, i.e. when no position moves) and votes on drops
    const float a4[4] = {S.x1[i], S.y1[i], S.x2[i], S.y2[i]};
    const float iarea = S.ar[i];
    const int first = i + 1, count = nd - first;
    bool any = false;
    unsigned long long nb = 0;
    for (int c0 = 0; c0 < count; c0 += kSoftThreads) {
      const int p = first + c0 + tid;
      if (p < nd) {
        const float b4[4] = {S.x1[p], S.y1[p], S.x2[p], S.y2[p]};
        const float ovr = hbb_overlap(a4, iarea, b4, S.ar[p]);
        float weight = 1.0f;
        if (method == 1) { if (ovr > thr) weight = 1.0f - ovr; }
        else if (method == 2) weight = soft_exp(-(ovr * ovr) / sigma);
        else weight = (ovr > thr) ? 0.0f : 1.0f;
        const float sn = weight * S.sc[p];
        S.sc[p] = sn;
        any = any || (sn < min_score);
        const unsigned long long k = pick_key(sn, p);
        nb = k > nb ? k : nb;
      }
    }
    nb = wave_max_key(nb);
    const bool wany = __any(any) != 0;
    if (lane == 0) { wbest[wave] = nb; wcnt[wave] = wany ? 1 : 0; }
    __syncthreads();
    unsigned long long g = wbest[0];
    int drops = wcnt[0];
#pragma unroll
    for (int k = 1; k < kW; ++k) { g = wbest[k] > g ? wbest[k] : g; drops |= wcnt[k]; }
    if (!drops) {                    // block-uniform: nothing moves, the vote above is the next pick
      if (first < nd) {
        const float s0 = S.sc[first];
        mp = (s0 != s0 || key_none(g)) ? first : key_pos(g);
      }
      continue;
    }
    __syncthreads();                 // wcnt is reused below
    // ---- (3) compaction of the drops, as the sequential loop would have left the array
    int drops_before = 0;
    int mydrops = 0;
    for (int c0 = 0; c0 < count; c0 += kSoftThreads) {
      const int p = first + c0 + tid;
      mydrops += (p < nd && S.sc[p] < min_score) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) mydrops += __shfl_xor(mydrops, o);
    if (lane == 0) wcnt[wave] = mydrops;
    __syncthreads();
    int D = 0;
#pragma unroll
    for (int k = 0; k < kW; ++k) D += wcnt[k];
    const int M = count - D, nd2 = first + M;
    __syncthreads();   // wcnt is reused below
    // holes: dropped positions below nd2, ranked ascending
    drops_before = 0;
    for (int c0 = 0; c0 < count; c0 += kSoftThreads) {
      const int p = first + c0 + tid;
      const bool drop = p < nd && S.sc[p] < min_score;
      const uint64_t mk = __ballot(drop);
      if (lane == 0) wcnt[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < kW; ++k) { const int v = wcnt[k]; off += (k < wave) ? v : 0; tot += v; }
      const int dpre = drops_before + off + __popcll(mk & ((1ull << lane) - 1ull));   // drops in (i, p)
      if (drop && p < nd2) hole[dpre] = p;
      drops_before += tot;
      __syncthreads();
    }
    // fillers: survivors at or above nd2; rank from the end = M - (survivors in (i, p])
    drops_before = 0;
    for (int c0 = 0; c0 < count; c0 += kSoftThreads) {
      const int p = first + c0 + tid;
      const bool drop = p < nd && S.sc[p] < min_score;
      const uint64_t mk = __ballot(drop);
      if (lane == 0) wcnt[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < kW; ++k) { const int v = wcnt[k]; off += (k < wave) ? v : 0; tot += v; }
      const int dpre = drops_before + off + __popcll(mk & ((1ull << lane) - 1ull));
      if (p < nd && !drop && p >= nd2) {
        const int kept_incl = (p - i) - dpre;          // survivors in (i, p], p itself included
        const int r = M - kept_incl;                   // survivors above p = rank from the end
        const int dst = hole[r];
        S.x1[dst] = S.x1[p]; S.y1[dst] = S.y1[p]; S.x2[dst] = S.x2[p]; S.y2[dst] = S.y2[p];
        S.sc[dst] = S.sc[p]; S.ar[dst] = S.ar[p]; S.id[dst] = S.id[p];
      }
      drops_before += tot;
      __syncthreads();
    }
    nd = nd2;
    if (first < nd) mp = argmax_from(first);   // positions moved: a pass of its own
  }
  __syncthreads();
  for (int p = tid; p < nd; p += kSoftThreads) {
    float *o = out_dets + (size_t)p * 5;
    o[0] = S.x1[p]; o[1] = S.y1[p]; o[2] = S.x2[p]; o[3] = S.y2[p]; o[4] = S.sc[p];
    out_inds[p] = S.id[p];
  }
  if (tid == 0) result[0] = nd;
}


// =====================================================================================================================
// Round 4: the state in REGISTERS, positions tracked instead of moved.
// The reference's array permutation is observable only through POSITIONS (who sits at i, who is first among equal
// scores, who fills a hole).  So a box never moves here: every thread keeps NB boxes in registers together with the
// position each of them currently has in the reference's array, and a round is
//   barrier -> read the waves' published candidates (key, coordinates, drop vote) -> the pick takes position i, the box
//   that sat there takes the pick's old position (two register compares per box, no LDS traffic) -> decay of the local
//   boxes against the pick's coordinates (boxes whose intersection is empty skip the division: their score is
//   unchanged bit for bit) -> the NEXT round's candidate of the wave (key by DPP moves, its coordinates, the vote on
//   drops) is published into the other half of a double buffer.
// ONE barrier among 4 waves per round without drops (the array version: four to six among 16).  A round WITH drops
// replays the reference's compaction on the positions: a bitmap of the dropped positions in LDS, its prefix counts,
// the table "k-th hole -> position", and the survivors at or above the new end take the holes from the end downwards
// (the same rule as in the array kernel above).  T = float | double (soft_nms_cpu_kernel<scalar_t>, nms_cpu.cpp:212-215:
// float64 input is computed in double; thresholds stay float and are promoted, :170,:184).
// =====================================================================================================================
struct SoftKey3 { uint32_t hi, mid, lo; };   // hi:mid = order-preserving image of the score (0 = NaN: never wins), lo = ~position
__device__ __forceinline__ bool key3_gt(const SoftKey3 &a, const SoftKey3 &b) {
  return a.hi > b.hi || (a.hi == b.hi && (a.mid > b.mid || (a.mid == b.mid && a.lo > b.lo)));
}
__device__ __forceinline__ void score_image(float s, uint32_t &hi, uint32_t &mid) {
  const float t = s + 0.0f;
  uint32_t u = __float_as_uint(t);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  hi = (s != s) ? 0u : u; mid = 0u;
}
__device__ __forceinline__ void score_image(double s, uint32_t &hi, uint32_t &mid) {
  const double t = s + 0.0;
  uint64_t u = (uint64_t)__double_as_longlong(t);
  u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
  if (s != s) u = 0ull;
  hi = (uint32_t)(u >> 32); mid = (uint32_t)u;
}
template <int CTRL>
__device__ __forceinline__ SoftKey3 dpp_max3(const SoftKey3 &v) {
  SoftKey3 o;
  o.hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.hi, CTRL, 0xf, 0xf, true);
  o.mid = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.mid, CTRL, 0xf, 0xf, true);
  o.lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.lo, CTRL, 0xf, 0xf, true);
  const bool gt = key3_gt(o, v);       // (component-wise selects: `gt ? o : v` on the structs goes through scratch)
  SoftKey3 r;
  r.hi = gt ? o.hi : v.hi; r.mid = gt ? o.mid : v.mid; r.lo = gt ? o.lo : v.lo;
  return r;
}
__device__ __forceinline__ SoftKey3 wave_max_key3(SoftKey3 v) {
  v = dpp_max3<0xB1>(v);
  v = dpp_max3<0x4E>(v);
  v = dpp_max3<0x141>(v);
  v = dpp_max3<0x140>(v);
  SoftKey3 m = {0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    SoftKey3 o;
    o.hi = (uint32_t)__builtin_amdgcn_readlane((int)v.hi, r * 16);
    o.mid = (uint32_t)__builtin_amdgcn_readlane((int)v.mid, r * 16);
    o.lo = (uint32_t)__builtin_amdgcn_readlane((int)v.lo, r * 16);
    const bool gt = key3_gt(o, m);
    m.hi = gt ? o.hi : m.hi; m.mid = gt ? o.mid : m.mid; m.lo = gt ? o.lo : m.lo;
  }
  return m;
}

constexpr int kSoftInf = 0x7FFFFFFF;      // position of a box that has left the array

template <class T, int THREADS, int NB>
__global__ __launch_bounds__(THREADS) void soft_nms_reg_kernel(const T *__restrict__ dets, int n, float thr_f, int method,
                                                               float sigma_f, float min_score_f, T *__restrict__ out_dets,
                                                               int64_t *__restrict__ out_inds, long long *__restrict__ result,
                                                               const int64_t *__restrict__ seg) {
  constexpr int kW = THREADS / 64, kCap = THREADS * NB;
  if (seg) {
    const int64_t off = seg[blockIdx.x];
    n = (int)(seg[blockIdx.x + 1] - off);
    dets += off * 5;
    out_dets += off * 5;
    out_inds += off;
    result += blockIdx.x;
    if (n <= 0) { if (threadIdx.x == 0) result[0] = 0; return; }          // (block-uniform)
    if (n > kCap) { if (threadIdx.x == 0) result[0] = -1; return; }        // (the host sizes the launch by the largest group)
  }
  __shared__ SoftKey3 pkey[2][kW];
  __shared__ T pcoord[2][kW][5];
  __shared__ int pdrop[2][kW];
  __shared__ uint32_t bitmap[kCap / 32], bpre[kCap / 32];
  __shared__ unsigned short holepos[kCap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T thr = (T)thr_f, sigma = (T)sigma_f, min_score = (T)min_score_f, zero = (T)0, one = (T)1;
  T x1[NB], y1[NB], x2[NB], y2[NB], sc[NB], ar[NB];
  int id[NB], pos[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int p = b * THREADS + tid;
    const bool in = p < n;
    const T *d = dets + (size_t)(in ? p : 0) * 5;
    x1[b] = d[0]; y1[b] = d[1]; x2[b] = d[2]; y2[b] = d[3]; sc[b] = in ? d[4] : zero;
    ar[b] = ((x2[b] - x1[b]) + one) * ((y2[b] - y1[b]) + one);
    id[b] = p;
    pos[b] = in ? p : kSoftInf;
  }
  int nd = n;
  // this wave's candidate for the pick among positions [from, nd): key by DPP, the owner lane publishes the coordinates
  auto publish = [&](int from, int par, bool anydrop) {
    SoftKey3 best = {0u, 0u, 0u};
    bool have = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (pos[b] >= from && pos[b] < nd) {
        SoftKey3 k;
        score_image(sc[b], k.hi, k.mid);
        // a NaN never wins (nms_cpu.cpp:116 `max_score < scores[pos]` is false for it) unless it sits at `from`, the
        // scan's starting value (:100-121)
        if (sc[b] != sc[b] && pos[b] == from) { k.hi = 0xFFFFFFFFu; k.mid = 0xFFFFFFFFu; }
        k.lo = ~(uint32_t)pos[b];
        const bool gt = !have || key3_gt(k, best);
        best.hi = gt ? k.hi : best.hi; best.mid = gt ? k.mid : best.mid; best.lo = gt ? k.lo : best.lo;
        have = true;
      }
    }
    if (!have) { best.hi = 0u; best.mid = 0u; best.lo = 0u; }     // (lo = 0 = position 2^32 - 1: below every real key)
    const SoftKey3 m = wave_max_key3(best);
    const bool wdrop = __any(anydrop) != 0;
    if (lane == 0) { pkey[par][wave] = m; pdrop[par][wave] = wdrop ? 1 : 0; }
    if (m.lo != 0u) {
      const int mpos = (int)(~m.lo);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (pos[b] == mpos) {
          pcoord[par][wave][0] = x1[b]; pcoord[par][wave][1] = y1[b]; pcoord[par][wave][2] = x2[b];
          pcoord[par][wave][3] = y2[b]; pcoord[par][wave][4] = ar[b];
        }
      }
    }
  };
  int par = 0;
  publish(0, par, false);
  for (int i = 0; i < nd; ++i) {
    __syncthreads();
    // ---- the winner among the waves' candidates
    SoftKey3 g = pkey[par][0];
    int gw = 0, drops = pdrop[par][0];
#pragma unroll
    for (int k = 1; k < kW; ++k) {
      const SoftKey3 o = pkey[par][k];
      drops |= pdrop[par][k];
      const bool gt = key3_gt(o, g);
      g.hi = gt ? o.hi : g.hi; g.mid = gt ? o.mid : g.mid; g.lo = gt ? o.lo : g.lo;
      gw = gt ? k : gw;
    }
    if (drops) {
      // ---- the previous round dropped boxes: compaction of the positions (first = i: the drops sit in [i, nd))
      // (1) bitmap of the dropped positions
      for (int w = tid; w < kCap / 32; w += THREADS) bitmap[w] = 0u;
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (pos[b] >= i && pos[b] < nd && sc[b] < min_score) atomicOr(&bitmap[pos[b] >> 5], 1u << (pos[b] & 31));
      __syncthreads();
      // (2) prefix counts per word (every wave computes them; wave 0 stores them)
      int D = 0;
      {
        int run = 0;
        for (int w0 = 0; w0 < kCap / 32; w0 += 64) {
          const bool inw = w0 + lane < kCap / 32;
          const int c = inw ? __popc(bitmap[w0 + lane]) : 0;
          int inc = c;
          for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
          if (wave == 0 && inw) bpre[w0 + lane] = (uint32_t)(run + inc - c);
          run += __shfl(inc, 63);
        }
        D = run;
      }
      const int count = nd - i, M = count - D, nd2 = i + M;
      __syncthreads();
      auto drops_before = [&](int p) -> int { return (int)bpre[p >> 5] + __popc(bitmap[p >> 5] & ((1u << (p & 31)) - 1u)); };
      // (3) the holes: dropped positions below the new end, ascending
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (pos[b] >= i && pos[b] < nd && sc[b] < min_score && pos[b] < nd2) holepos[drops_before(pos[b])] = (unsigned short)pos[b];
      __syncthreads();
      // (4) survivors at or above the new end fill the holes from the END downwards; dropped boxes leave
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (pos[b] >= i && pos[b] < nd) {
          if (sc[b] < min_score) pos[b] = kSoftInf;
          else if (pos[b] >= nd2) {
            const int kept_incl = (pos[b] - (i - 1)) - drops_before(pos[b]);   // survivors in [i, p], p itself included
            pos[b] = (int)holepos[M - kept_incl];                            // survivors above p = rank from the end
          }
        }
      }
      nd = nd2;
      par ^= 1;
      publish(i, par, false);         // positions moved: the candidates are taken again (no drop vote this time)
      --i;                            // (this iteration placed nothing)
      continue;
    }
    // ---- the pick takes position i; the box that sat there takes the pick's old position
    const int mp = g.lo != 0u ? (int)(~g.lo) : i;
    const T a0 = pcoord[par][gw][0], a1 = pcoord[par][gw][1], a2 = pcoord[par][gw][2], a3 = pcoord[par][gw][3], iarea = pcoord[par][gw][4];
    bool anydrop = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (pos[b] == mp) pos[b] = i;
      else if (pos[b] == i) pos[b] = mp;
      if (pos[b] > i && pos[b] < nd) {
        // nms_cpu.cpp:155-163 (= hbb_overlap) with the pick as box i
        const T xx1 = a0 < x1[b] ? x1[b] : a0, yy1 = a1 < y1[b] ? y1[b] : a1;
        const T xx2 = x2[b] < a2 ? x2[b] : a2, yy2 = y2[b] < a3 ? y2[b] : a3;
        const T ww = (xx2 - xx1) + one, hh = (yy2 - yy1) + one;
        const T w = zero < ww ? ww : zero, h = zero < hh ? hh : zero;
        const T inter = w * h;
        const T den = (iarea + ar[b]) - inter;
        // empty intersection over a positive denominator: ovr = +0, every method's weight is exactly 1 (hard / linear:
        // `0 > thr` must be false), the score is unchanged bit for bit -- no division, no exp
        const bool skip = inter == zero && den > zero && (method == 2 || !(zero > thr));
        if (!skip) {
          const T ovr = inter / den;
          T weight = one;
          if (method == 1) { if (ovr > thr) weight = one - ovr; }
          else if (method == 2) weight = soft_exp(-(ovr * ovr) / sigma);
          else weight = (ovr > thr) ? zero : one;
          sc[b] = weight * sc[b];
        }
        anydrop = anydrop || (sc[b] < min_score);
      }
    }
    par ^= 1;
    publish(i + 1, par, anydrop);
  }
  // (no drop vote can be pending here: a round that dropped boxes is followed by one more iteration -- i < nd still
  // held -- whose compaction branch applied them; the round of the last position decays nothing)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (pos[b] < nd) {
      T *o = out_dets + (size_t)pos[b] * 5;
      o[0] = x1[b]; o[1] = y1[b]; o[2] = x2[b]; o[3] = y2[b]; o[4] = sc[b];
      out_inds[pos[b]] = id[b];
    }
  }
  if (tid == 0) result[0] = nd;
}

// =====================================================================================================================
// Round 4, second attempt at one array: the reference's array permutation is observable only where two REMAINING boxes
// hold the same score at the moment one of them is the maximum (the strict `<` of the arg-max then prefers the lower
// POSITION, nms_cpu.cpp:114-121) -- or where a score is NaN.  Everywhere else the pick of a round is the unique maximum,
// whoever sits where, every remaining box is examined exactly once per round (a box pulled into a hole is examined
// there, :184-198), and the output is the picks in order.  So this kernel drops the positions altogether: a box stays
// in its register slot with a `live` bit, and a round is
//     barrier -> the block's pick from two LDS words (the best score with its lowest and with its highest box number:
//     they differ when the maximum is not unique) -> the pick's corners from a read-only LDS copy -> decay of the live
//     boxes -- skipped by a whole wave when all its boxes lie left or right of the pick: the boxes are placed in x order,
//     64 NB consecutive ranks per wave -- and the drop test on every live box, as in the reference -> a wave whose
//     candidate was picked, decayed or dropped finds its next one by DPP moves; every wave feeds its candidate to the
//     next round's two LDS maxima.
// ONE barrier per round.  Measured (tools/bench_soft_nms.py, boxes of 8-200 px on a 900 px canvas): n = 2000 2.0 ms
// against 3.3 ms of the array kernel and 3.2 ms of the host loop; 4096 boxes: 6.0 against 11.4 and 14.4 ms -- the same on
// sparse scenes: a round is ~1 us whatever the boxes do.  What a round waits for is ONE wave's dependent chain (stamps:
// tools/exp_soft_stamps.py): barrier -> LDS -> LDS -> (division) -> score image, four DPP steps, read-lanes, ballots of
// the wave that owned the pick -> LDS atomics -> barrier; trimming the other waves' instructions (the candidate reduction
// by every wave: -30 instructions; no republishing unless touched: -50; no decay away from the pick: -100) moved it
// from 1.1 to 1.0 us.  Fewer rounds (several non-interacting picks per barrier) would be the next step; not built.  A
// NaN score, or a maximum that two remaining boxes share, ends the kernel with result -2 and the call is repeated on the
// array kernel, which replays the permutation exactly.
// =====================================================================================================================
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = dpp_max_u32<0xB1>(v);
  v = dpp_max_u32<0x4E>(v);
  v = dpp_max_u32<0x141>(v);
  v = dpp_max_u32<0x140>(v);
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, r * 16);
    m = o > m ? o : m;
  }
  return m;
}
// order-preserving image of a score (-0 folded into +0): > 0 for every score
// (a NaN -- a decayed score can become one through NaN corners -- maps to the top value: it surfaces as the next maximum
// and the kernel gives up there)
__device__ __forceinline__ uint32_t score_image32(float s) {
  const uint32_t u = __float_as_uint(s + 0.0f);
  return s != s ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}

constexpr long long kSoftFastGiveUp = -2;
#ifdef OBB_SOFT_STAMPS
__device__ unsigned long long g_soft_stamps[16 * 256 * 8];
#define SF_STAMP(i) do { if (lane == 0 && picks >= 256 && picks < 512) g_soft_stamps[(wave * 256 + (picks - 256)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SF_STAMP(i) do { } while (0)
#endif

template <int T, int NB>
__global__ __launch_bounds__(T) void soft_nms_fast_kernel(const float *__restrict__ dets, int n, float thr, int method,
                                                            float sigma, float min_score, float *__restrict__ out_dets,
                                                            int64_t *__restrict__ out_inds, long long *__restrict__ result,
                                                            const int64_t *__restrict__ seg) {
  constexpr int kW = T / 64;
  static_assert(kW <= 16, "one row of 16 lanes reduces the waves' candidates");
  if (seg) {      // GROUPED (as soft_nms_kernel): workgroup g on the rows [seg[g], seg[g + 1]) as an array of its own
    const int64_t off = seg[blockIdx.x];
    const int64_t m = seg[blockIdx.x + 1] - off;
    dets += off * 5;
    out_dets += off * 5;
    out_inds += off;
    result += blockIdx.x;
    if (m <= 0) { if (threadIdx.x == 0) result[0] = 0; return; }                        // (block-uniform)
    if (m > T * NB) { if (threadIdx.x == 0) result[0] = kSoftFastGiveUp; return; }      // (beyond the caller's bound: the array kernel's case)
    n = (int)m;
  }
  static_assert((NB & (NB - 1)) == 0, "slot = (box number >> 6) & (NB - 1)");
  __shared__ float4 cbox[T * NB];
  __shared__ float carea[T * NB];
  __shared__ uint32_t ckey[T * NB];         // order-preserving images of x1 while the boxes are ranked, then the scores
  __shared__ int cid[T * NB];
  // the block's pick of a round = two LDS maxima the waves feed before the barrier (three sets in rotation):
  //   [0] (score image << 32) | ~box number  -- the best score, its LOWEST box number
  //   [1] (score image << 32) |  box number  -- the best score, its HIGHEST box number; the two differ: not unique
  __shared__ unsigned long long gbest[3][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float x1[NB], y1[NB], x2[NB], y2[NB], sc[NB], ar[NB];
  int rank[NB], oid[NB];
  uint32_t live = 0u;
  bool bad = false;
  // ---- the boxes are placed in x ORDER: wave w holds the boxes of ranks [w 64 NB, (w + 1) 64 NB) by (x1, row) -- a pick
  // then meets the columns of a few waves only and the others skip the decay on a wave-uniform test (below).  Nothing
  // observable depends on the placement: this kernel has no positions.
  {
    uint32_t key[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int p = b * T + tid;
      const bool in = p < n;
      const float *d = dets + (size_t)(in ? p : 0) * 5;
      x1[b] = d[0]; y1[b] = d[1]; x2[b] = d[2]; y2[b] = d[3]; sc[b] = d[4];
      key[b] = in ? score_image32(x1[b]) : 0xFFFFFFFFu;           // (NaN: the top value, like the absent rows)
      if (in) bad = bad || (sc[b] != sc[b]);
      ckey[p] = key[b];
    }
    if (__syncthreads_or(bad ? 1 : 0)) {       // (also publishes the keys)
      if (tid == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    int r[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) r[b] = 0;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(ckey);
    for (int q = 0; q < T * NB; q += 4) {        // (uniform addresses: LDS broadcast reads)
      const uint4 k = k4[q >> 2];
#pragma unroll
      for (int b = 0; b < NB; ++b) {           // (key, row) as one 64-bit number: a strict total order
        const unsigned long long me = ((unsigned long long)key[b] << 32) | (uint32_t)(b * T + tid);
        r[b] += ((((unsigned long long)k.x << 32) | (uint32_t)q) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.y << 32) | (uint32_t)(q + 1)) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.z << 32) | (uint32_t)(q + 2)) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.w << 32) | (uint32_t)(q + 3)) < me) ? 1 : 0;
      }
    }
    __syncthreads();                             // every thread is done with the keys
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int p = b * T + tid;
      cbox[r[b]] = make_float4(x1[b], y1[b], x2[b], y2[b]);
      ckey[r[b]] = __float_as_uint(sc[b]);
      cid[r[b]] = p < n ? p : -1;
    }
    __syncthreads();
  }
  float wxlo = 3.0e38f, wxhi = -3.0e38f, war_min = 3.0e38f;
  bool irregular = false;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int spos = wave * (64 * NB) + b * 64 + lane;            // the box number used from here on
    const float4 c = cbox[spos];
    x1[b] = c.x; y1[b] = c.y; x2[b] = c.z; y2[b] = c.w; sc[b] = __uint_as_float(ckey[spos]);
    oid[b] = cid[spos];
    ar[b] = ((x2[b] - x1[b]) + 1.0f) * ((y2[b] - y1[b]) + 1.0f);
    carea[spos] = ar[b];
    rank[b] = -1;
    if (oid[b] >= 0) {
      live |= 1u << b;
      const float big = 3.0e38f;
      irregular = irregular || !(fabsf(x1[b]) < big && fabsf(y1[b]) < big && fabsf(x2[b]) < big && fabsf(y2[b]) < big && fabsf(ar[b]) < big);
      wxlo = x1[b] < wxlo ? x1[b] : wxlo; wxhi = x2[b] > wxhi ? x2[b] : wxhi;
      war_min = ar[b] < war_min ? ar[b] : war_min;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float l = __shfl_xor(wxlo, o), h = __shfl_xor(wxhi, o), m = __shfl_xor(war_min, o);
    wxlo = l < wxlo ? l : wxlo; wxhi = h > wxhi ? h : wxhi; war_min = m < war_min ? m : war_min;
  }
  const bool wave_regular = __ballot(irregular) == 0ull;          // every corner and area of the wave's boxes finite
  __syncthreads();                               // (carea)
  // the wave's candidate: {score image (0: none), box number of its first owner, of its last owner}; wave-uniform
  auto candidate = [&]() {
    uint32_t img[NB], best = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      img[b] = (live >> b & 1u) ? score_image32(sc[b]) : 0u;
      best = img[b] > best ? img[b] : best;
    }
    const uint32_t wmax = wave_max_u32(best);
    uint32_t eq = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) eq |= (img[b] == wmax && wmax != 0u) ? (1u << b) : 0u;
    const uint64_t owners = __ballot(eq != 0u);
    uint32_t lo = 0u, hi = 0u;
    if (owners) {
      const int fl = __builtin_ctzll(owners), ll = 63 - __builtin_clzll(owners);
      const uint32_t feq = (uint32_t)__builtin_amdgcn_readlane((int)eq, fl), leq = (uint32_t)__builtin_amdgcn_readlane((int)eq, ll);
      lo = (uint32_t)(wave * (64 * NB) + __builtin_ctz(feq) * 64 + fl);
      hi = (uint32_t)(wave * (64 * NB) + (31 - __builtin_clz(leq)) * 64 + ll);
    }
    return make_uint3(wmax, lo, hi);
  };
  auto feed = [&](int set, const uint3 &c) {
    if (lane == 0 && c.x != 0u) {
      atomicMax(&gbest[set][0], ((unsigned long long)c.x << 32) | (uint32_t)~c.y);
      atomicMax(&gbest[set][1], ((unsigned long long)c.x << 32) | c.z);
    }
  };
  if (tid < 6) (&gbest[0][0])[tid] = 0ull;
  __syncthreads();
  int par = 0, picks = 0;                    // par: the set of this round (0, 1, 2, 0, ...)
  uint3 wc = candidate();
  feed(par, wc);
  for (;;) {
    SF_STAMP(0);
    __syncthreads();
    SF_STAMP(1);
    const unsigned long long ga = gbest[par][0], gb = gbest[par][1];      // (uniform addresses)
    const int nxt = par == 2 ? 0 : par + 1, nn = nxt == 2 ? 0 : nxt + 1;
    if (tid == 0) { gbest[nn][0] = 0ull; gbest[nn][1] = 0ull; }          // the set after the next one: last read two barriers ago
    const uint32_t gmax = (uint32_t)(ga >> 32);
    if (gmax == 0u) break;                                       // nothing left
    const int gi = (int)(~(uint32_t)ga);
    if (gi != (int)(uint32_t)gb || gmax == 0xFFFFFFFFu) {        // (block-uniform) the maximum is not unique, or a NaN
      if (tid == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    SF_STAMP(2);
    const float4 a = cbox[gi];
    const float iarea = carea[gi];
    const bool mine = (gi & 63) == lane && gi / (64 * NB) == wave;
    SF_STAMP(3);
    const int gslot = (gi >> 6) & (NB - 1);
    // the wave's candidate stays what it is unless its box is picked, decayed or dropped in this round: every other box
    // of the wave was below it and only goes down (a "not unique" mark may outlive the tie: the kernel then gives up
    // without need, never the other way round)
    const int wbest = (int)wc.y;
    const bool own_best = wc.x != 0u && (wbest & 63) == lane;
    const int wslot = (wbest >> 6) & (NB - 1);
    bool touched = false;
    if (mine) {
#pragma unroll
      for (int b = 0; b < NB; ++b) if (b == gslot) { live &= ~(1u << b); rank[b] = picks; touched = true; }
    }
    // a wave all of whose boxes lie left or right of the pick (x order) skips the decay: for each of them w = 0, h is
    // finite, the intersection +0, the denominator iarea + area > 0 -- the case the per-box test below calls `skip`
    const float big = 3.0e38f;
    const bool plain = method == 2 || !(0.0f > thr);
    const bool far = wave_regular && plain && fabsf(a.y) < big && fabsf(a.w) < big && (iarea + war_min) > 0.0f &&
                     (a.z + 2.0f < wxlo || a.x - 2.0f > wxhi);
    if (!far) {
      // nms_cpu.cpp:155-163 (= hbb_overlap) with the pick as box i, for all the lane's boxes first ...
      float inter[NB], den[NB];
      bool act[NB];
      bool any = false;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t bit = 1u << b;
        const float xx1 = a.x < x1[b] ? x1[b] : a.x, yy1 = a.y < y1[b] ? y1[b] : a.y;
        const float xx2 = x2[b] < a.z ? x2[b] : a.z, yy2 = y2[b] < a.w ? y2[b] : a.w;
        const float ww = (xx2 - xx1) + 1.0f, hh = (yy2 - yy1) + 1.0f;
        const float w = 0.0f < ww ? ww : 0.0f, h = 0.0f < hh ? hh : 0.0f;
        inter[b] = w * h;
        den[b] = (iarea + ar[b]) - inter[b];
        // empty intersection over a positive denominator: ovr = +0, every method's weight is exactly 1 (hard / linear:
        // `0 > thr` must be false), the score is unchanged bit for bit -- no division, no exp
        const bool skip = inter[b] == 0.0f && den[b] > 0.0f && (method == 2 || !(0.0f > thr));
        act[b] = (live & bit) != 0u && !skip;
        any = any || act[b];
      }
      // ... then ONE region with the divisions of all of them (a branch per box kept their dependent chains -- IEEE
      // division, the exponential -- one behind the other: the decay was the longest leg of a round by the stamps)
      if (__ballot(any) != 0ull) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float ovr = inter[b] / den[b];
          float weight = 1.0f;
          if (method == 1) { if (ovr > thr) weight = 1.0f - ovr; }
          else if (method == 2) weight = soft_exp(-(ovr * ovr) / sigma);
          else weight = (ovr > thr) ? 0.0f : 1.0f;
          if (act[b]) {
            sc[b] = weight * sc[b];
            touched = touched || (own_best && b == wslot);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t bit = 1u << b;
      if ((live & bit) != 0u && sc[b] < min_score) { live &= ~bit; touched = touched || (own_best && b == wslot); }   // (every examined box, decayed or not: nms_cpu.cpp:184)
    }
    SF_STAMP(4);
    ++picks;
    par = nxt;
    if (__ballot(touched) != 0ull) wc = candidate();             // (wave-uniform)
    feed(par, wc);
    --picks; SF_STAMP(5); ++picks;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (rank[b] >= 0) {
      float *o = out_dets + (size_t)rank[b] * 5;
      o[0] = x1[b]; o[1] = y1[b]; o[2] = x2[b]; o[3] = y2[b]; o[4] = sc[b];
      out_inds[rank[b]] = oid[b];
    }
  }
  if (tid == 0) result[0] = picks;
}

// ---- the position-free kernel on ONE wave (round 5): small arrays -- the per-class arrays of a detection head -- need no
// block: NB <= 8 boxes per lane, the round's pick is a wave reduction (four DPP steps, four
// read-lanes) instead of two LDS maxima behind a barrier, the pick's corners come from a read-only LDS copy by a
// broadcast read; no barrier, no atomics.  The rounds, the decay arithmetic and the give-up rule (a maximum shared by two
// remaining boxes, a NaN score: result -2, the array kernel replays the call) are soft_nms_fast_kernel's -- same picks,
// same bits.  A grouped call gives every group a wave of its own (one block of 64 threads per group).
template <int NB>
__global__ __launch_bounds__(64) void soft_nms_wave_kernel(const float *__restrict__ dets, int n, float thr, int method,
                                                           float sigma, float min_score, float *__restrict__ out_dets,
                                                           int64_t *__restrict__ out_inds, long long *__restrict__ result,
                                                           const int64_t *__restrict__ seg) {
  if (seg) {
    const int64_t off = seg[blockIdx.x];
    const int64_t m = seg[blockIdx.x + 1] - off;
    dets += off * 5;
    out_dets += off * 5;
    out_inds += off;
    result += blockIdx.x;
    if (m <= 0) { if (threadIdx.x == 0) result[0] = 0; return; }
    if (m > 64 * NB) { if (threadIdx.x == 0) result[0] = kSoftFastGiveUp; return; }     // (beyond the caller's bound)
    n = (int)m;
  }
  __shared__ float4 cbox[64 * NB];
  __shared__ float carea[64 * NB];
  const int lane = threadIdx.x;
  float x1[NB], y1[NB], x2[NB], y2[NB], sc[NB], ar[NB];
  int rank[NB];
  uint32_t live = 0u;
  bool bad = false;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int p = b * 64 + lane;                 // the box number: slot b of lane `lane`
    const bool in = p < n;
    const float *d = dets + (size_t)(in ? p : 0) * 5;
    x1[b] = d[0]; y1[b] = d[1]; x2[b] = d[2]; y2[b] = d[3]; sc[b] = d[4];
    ar[b] = ((x2[b] - x1[b]) + 1.0f) * ((y2[b] - y1[b]) + 1.0f);
    rank[b] = -1;
    if (in) { live |= 1u << b; bad = bad || (sc[b] != sc[b]); }
    cbox[p] = make_float4(x1[b], y1[b], x2[b], y2[b]);
    carea[p] = ar[b];
  }
  if (__ballot(bad) != 0ull) { if (lane == 0) result[0] = kSoftFastGiveUp; return; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int picks = 0;
  for (;;) {
    // the round's pick: the unique maximum of the live scores
    uint32_t img[NB], best = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      img[b] = (live >> b & 1u) ? score_image32(sc[b]) : 0u;
      best = img[b] > best ? img[b] : best;
    }
    const uint32_t wmax = wave_max_u32(best);
    if (wmax == 0u) break;                                       // nothing left
    uint32_t eq = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) eq |= (img[b] == wmax) ? (1u << b) : 0u;
    const uint64_t owners = __ballot(eq != 0u);
    const int fl = __builtin_ctzll(owners);
    const uint32_t feq = (uint32_t)__builtin_amdgcn_readlane((int)eq, fl);
    if (wmax == 0xFFFFFFFFu || (owners & (owners - 1ull)) != 0ull || (feq & (feq - 1u)) != 0u) {   // a NaN, or the maximum is not unique
      if (lane == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    const int fs = __builtin_ctz(feq), gi = fs * 64 + fl;
    const float4 a = cbox[gi];
    const float iarea = carea[gi];
    if (lane == fl) {
#pragma unroll
      for (int b = 0; b < NB; ++b) if (b == fs) { live &= ~(1u << b); rank[b] = picks; }
    }
    // nms_cpu.cpp:155-163 (= hbb_overlap) with the pick as box i, for all the lane's boxes first ...
    float inter[NB], den[NB];
    bool act[NB];
    bool any = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t bit = 1u << b;
      const float xx1 = a.x < x1[b] ? x1[b] : a.x, yy1 = a.y < y1[b] ? y1[b] : a.y;
      const float xx2 = x2[b] < a.z ? x2[b] : a.z, yy2 = y2[b] < a.w ? y2[b] : a.w;
      const float ww = (xx2 - xx1) + 1.0f, hh = (yy2 - yy1) + 1.0f;
      const float w = 0.0f < ww ? ww : 0.0f, h = 0.0f < hh ? hh : 0.0f;
      inter[b] = w * h;
      den[b] = (iarea + ar[b]) - inter[b];
      // empty intersection over a positive denominator: ovr = +0, every method's weight is exactly 1: unchanged bit for bit
      const bool skip = inter[b] == 0.0f && den[b] > 0.0f && (method == 2 || !(0.0f > thr));
      act[b] = (live & bit) != 0u && !skip;
      any = any || act[b];
    }
    // ... then ONE region with the divisions of all of them
    if (__ballot(any) != 0ull) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float ovr = inter[b] / den[b];
        float weight = 1.0f;
        if (method == 1) { if (ovr > thr) weight = 1.0f - ovr; }
        else if (method == 2) weight = soft_exp(-(ovr * ovr) / sigma);
        else weight = (ovr > thr) ? 0.0f : 1.0f;
        if (act[b]) sc[b] = weight * sc[b];
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t bit = 1u << b;
      if ((live & bit) != 0u && sc[b] < min_score) live &= ~bit;      // (every examined box, decayed or not: nms_cpu.cpp:184)
    }
    ++picks;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (rank[b] >= 0) {
      float *o = out_dets + (size_t)rank[b] * 5;
      o[0] = x1[b]; o[1] = y1[b]; o[2] = x2[b]; o[3] = y2[b]; o[4] = sc[b];
      out_inds[rank[b]] = b * 64 + lane;
    }
  }
  if (lane == 0) result[0] = picks;
}

// ---- the position-free kernel for float64 rows (soft_nms_cpu_kernel<double>): the same rounds, but a double's order image is
// 64 bits wide and does not share a word with a box number -- the block's pick takes TWO LDS steps (the maximum image, then
// the lowest / highest box number among the waves that hold it) and two barriers per round.  Up to 2048 rows (56 bytes of
// LDS per box); shared maxima / NaN scores give up like the float kernel and the register kernel replays the call.
__device__ __forceinline__ unsigned long long score_image64(double s) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(s + 0.0);
  return s != s ? ~0ull : ((u >> 63) ? ~u : (u | (1ull << 63)));
}

template <int T, int NB>
__global__ __launch_bounds__(T) void soft_nms_fast64_kernel(const double *__restrict__ dets, int n, float thr_f, int method,
                                                              float sigma_f, float min_score_f, double *__restrict__ out_dets,
                                                              int64_t *__restrict__ out_inds, long long *__restrict__ result) {
  static_assert((NB & (NB - 1)) == 0, "slot = (box number >> 6) & (NB - 1)");
  __shared__ double4 cbox[T * NB];
  __shared__ double carea[T * NB], cscore[T * NB];
  __shared__ uint32_t ckey[T * NB];         // order-preserving images of (float)x1: the placement only
  __shared__ int cid[T * NB];
  __shared__ unsigned long long gmax[2];     // the best score image of a round (two sets in turn)
  __shared__ uint32_t glo[2], ghi[2];        // ... the lowest / highest box number that holds it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double thr = (double)thr_f, sigma = (double)sigma_f, min_score = (double)min_score_f;     // (promoted as in the double instantiation, nms_cpu.cpp:170,184)
  double x1[NB], y1[NB], x2[NB], y2[NB], sc[NB], ar[NB];
  int rank[NB], oid[NB];
  uint32_t live = 0u;
  bool bad = false;
  {
    uint32_t key[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int p = b * T + tid;
      const bool in = p < n;
      const double *d = dets + (size_t)(in ? p : 0) * 5;
      x1[b] = d[0]; y1[b] = d[1]; x2[b] = d[2]; y2[b] = d[3]; sc[b] = d[4];
      key[b] = in ? score_image32((float)x1[b]) : 0xFFFFFFFFu;
      if (in) bad = bad || (sc[b] != sc[b]);
      ckey[p] = key[b];
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
      if (tid == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    int r[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) r[b] = 0;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(ckey);
    for (int q = 0; q < T * NB; q += 4) {
      const uint4 k = k4[q >> 2];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const unsigned long long me = ((unsigned long long)key[b] << 32) | (uint32_t)(b * T + tid);
        r[b] += ((((unsigned long long)k.x << 32) | (uint32_t)q) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.y << 32) | (uint32_t)(q + 1)) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.z << 32) | (uint32_t)(q + 2)) < me) ? 1 : 0;
        r[b] += ((((unsigned long long)k.w << 32) | (uint32_t)(q + 3)) < me) ? 1 : 0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int p = b * T + tid;
      cbox[r[b]] = make_double4(x1[b], y1[b], x2[b], y2[b]);
      cscore[r[b]] = sc[b];
      cid[r[b]] = p < n ? p : -1;
    }
    __syncthreads();
  }
  const double big = 1.0e308;
  double wxlo = big, wxhi = -big, war_min = big;
  bool irregular = false;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int spos = wave * (64 * NB) + b * 64 + lane;
    const double4 c = cbox[spos];
    x1[b] = c.x; y1[b] = c.y; x2[b] = c.z; y2[b] = c.w; sc[b] = cscore[spos];
    oid[b] = cid[spos];
    ar[b] = ((x2[b] - x1[b]) + 1.0) * ((y2[b] - y1[b]) + 1.0);
    carea[spos] = ar[b];
    rank[b] = -1;
    if (oid[b] >= 0) {
      live |= 1u << b;
      irregular = irregular || !(fabs(x1[b]) < big && fabs(y1[b]) < big && fabs(x2[b]) < big && fabs(y2[b]) < big && fabs(ar[b]) < big);
      wxlo = x1[b] < wxlo ? x1[b] : wxlo; wxhi = x2[b] > wxhi ? x2[b] : wxhi;
      war_min = ar[b] < war_min ? ar[b] : war_min;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double l = __shfl_xor(wxlo, o), h = __shfl_xor(wxhi, o), m = __shfl_xor(war_min, o);
    wxlo = l < wxlo ? l : wxlo; wxhi = h > wxhi ? h : wxhi; war_min = m < war_min ? m : war_min;
  }
  const bool wave_regular = __ballot(irregular) == 0ull;
  unsigned long long cw = 0ull;              // the wave's candidate: score image (0: none), first / last owner's box number
  uint32_t clo = 0u, chi = 0u;
  auto candidate = [&]() __attribute__((always_inline)) {
    unsigned long long img[NB], best = 0ull;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      img[b] = (live >> b & 1u) ? score_image64(sc[b]) : 0ull;
      best = img[b] > best ? img[b] : best;
    }
    const unsigned long long wmax = wave_max_key(best);
    uint32_t eq = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) eq |= (img[b] == wmax && wmax != 0ull) ? (1u << b) : 0u;
    const uint64_t owners = __ballot(eq != 0u);
    clo = 0u; chi = 0u;
    if (owners) {
      const int fl = __builtin_ctzll(owners), ll = 63 - __builtin_clzll(owners);
      const uint32_t feq = (uint32_t)__builtin_amdgcn_readlane((int)eq, fl), leq = (uint32_t)__builtin_amdgcn_readlane((int)eq, ll);
      clo = (uint32_t)(wave * (64 * NB) + __builtin_ctz(feq) * 64 + fl);
      chi = (uint32_t)(wave * (64 * NB) + (31 - __builtin_clz(leq)) * 64 + ll);
    }
    cw = wmax;
  };
  if (tid < 2) { gmax[tid] = 0ull; glo[tid] = 0xFFFFFFFFu; ghi[tid] = 0u; }
  __syncthreads();
  int par = 0, picks = 0;
  candidate();
  if (lane == 0 && cw != 0ull) atomicMax(&gmax[par], cw);
  for (;;) {
    __syncthreads();                                             // the round's maximum image is complete
    const unsigned long long gm = gmax[par];
    const int nxt = par ^ 1;
    if (tid == 0) { gmax[nxt] = 0ull; glo[nxt] = 0xFFFFFFFFu; ghi[nxt] = 0u; }     // (last read before this barrier, fed after the next one)
    if (gm == 0ull) break;                                       // nothing left
    if (gm == ~0ull) {                                           // (block-uniform) a NaN score
      if (tid == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    if (lane == 0 && cw == gm) { atomicMin(&glo[par], clo); atomicMax(&ghi[par], chi); }
    __syncthreads();                                             // ... and who holds it
    const int gi = (int)glo[par];
    if (glo[par] != ghi[par]) {                                  // (block-uniform) the maximum is not unique
      if (tid == 0) result[0] = kSoftFastGiveUp;
      return;
    }
    const double4 a = cbox[gi];
    const double iarea = carea[gi];
    const bool mine = (gi & 63) == lane && gi / (64 * NB) == wave;
    const int gslot = (gi >> 6) & (NB - 1);
    const int wbest = (int)clo;
    const bool own_best = cw != 0ull && (wbest & 63) == lane;
    const int wslot = (wbest >> 6) & (NB - 1);
    bool touched = false;
    if (mine) {
#pragma unroll
      for (int b = 0; b < NB; ++b) if (b == gslot) { live &= ~(1u << b); rank[b] = picks; touched = true; }
    }
    const bool plain = method == 2 || !(0.0 > thr);
    const bool far = wave_regular && plain && fabs(a.y) < big && fabs(a.w) < big && (iarea + war_min) > 0.0 &&
                     (a.z + 2.0 < wxlo || a.x - 2.0 > wxhi);
    if (!far) {
      double inter[NB], den[NB];
      bool act[NB];
      bool any = false;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t bit = 1u << b;
        const double xx1 = a.x < x1[b] ? x1[b] : a.x, yy1 = a.y < y1[b] ? y1[b] : a.y;
        const double xx2 = x2[b] < a.z ? x2[b] : a.z, yy2 = y2[b] < a.w ? y2[b] : a.w;
        const double ww = (xx2 - xx1) + 1.0, hh = (yy2 - yy1) + 1.0;
        const double w = 0.0 < ww ? ww : 0.0, h = 0.0 < hh ? hh : 0.0;
        inter[b] = w * h;
        den[b] = (iarea + ar[b]) - inter[b];
        const bool skip = inter[b] == 0.0 && den[b] > 0.0 && plain;
        act[b] = (live & bit) != 0u && !skip;
        any = any || act[b];
      }
      if (__ballot(any) != 0ull) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const double ovr = inter[b] / den[b];
          double weight = 1.0;
          if (method == 1) { if (ovr > thr) weight = 1.0 - ovr; }
          else if (method == 2) weight = soft_exp(-(ovr * ovr) / sigma);
          else weight = (ovr > thr) ? 0.0 : 1.0;
          if (act[b]) {
            sc[b] = weight * sc[b];
            touched = touched || (own_best && b == wslot);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t bit = 1u << b;
      if ((live & bit) != 0u && sc[b] < min_score) { live &= ~bit; touched = touched || (own_best && b == wslot); }
    }
    ++picks;
    par = nxt;
    if (__ballot(touched) != 0ull) candidate();                  // (wave-uniform)
    if (lane == 0 && cw != 0ull) atomicMax(&gmax[par], cw);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (rank[b] >= 0) {
      double *o = out_dets + (size_t)rank[b] * 5;
      o[0] = x1[b]; o[1] = y1[b]; o[2] = x2[b]; o[3] = y2[b]; o[4] = sc[b];
      out_inds[rank[b]] = oid[b];
    }
  }
  if (tid == 0) result[0] = picks;
}

constexpr int kSoftFast64Max = 2048;
static bool soft_launch_fast64(const double *dets, int64_t n, float thr, int method, float sigma, float min_score,
                               double *out_dets, int64_t *out_inds, long long *result, hipStream_t st) {
#define OBB_SOFT_FAST64(T, NB) hipLaunchKernelGGL((soft_nms_fast64_kernel<T, NB>), dim3(1), dim3(T), 0, st, dets, (int)n, thr, method, sigma, min_score, out_dets, out_inds, result)
  if (n <= 256) OBB_SOFT_FAST64(256, 1);
  else if (n <= 512) OBB_SOFT_FAST64(512, 1);
  else if (n <= 1024) OBB_SOFT_FAST64(1024, 1);
  else if (n <= kSoftFast64Max) OBB_SOFT_FAST64(1024, 2);
  else return false;
#undef OBB_SOFT_FAST64
  return true;
}

static thread_local int g_soft_path = -1;      // obb_soft_nms_last_path
constexpr int kSoftFastMax = 4096;
constexpr int kSoftWaveMax = 64;                // soft_nms_wave_kernel by default: one box per lane (see soft_launch_fast)
// (n = the array's rows, or -- grouped, seg != nullptr -- the caller's bound on the rows of one group)
static bool soft_launch_fast(const float *dets, int64_t n, float thr, int method, float sigma, float min_score,
                             float *out_dets, int64_t *out_inds, long long *result, hipStream_t st,
                             const int64_t *seg = nullptr, unsigned groups = 1) {
#define OBB_SOFT_FAST(T, NB) hipLaunchKernelGGL((soft_nms_fast_kernel<T, NB>), dim3(groups), dim3(T), 0, st, dets, (int)n, thr, method, sigma, min_score, out_dets, out_inds, result, seg)
  // (four waves per SIMD hide the latency of the dependent chain -- division, the exponential -- that one wave per SIMD
  // waits out: 2048 boxes as 256 x 8 took 2.1 us per round in a dense scene)
  const int64_t shape = tuning(OBB_TUNE_SOFT_NMS_SHAPE, 0);     // (0 default; 1: 256 threads x more boxes, for measurements; 3: never the one-wave kernel)
#define OBB_SOFT_WAVE(NB) hipLaunchKernelGGL((soft_nms_wave_kernel<NB>), dim3(groups), dim3(64), 0, st, dets, (int)n, thr, method, sigma, min_score, out_dets, out_inds, result, seg)
  // One wave serves the arrays of up to 64 boxes (one per lane): 45 us per call against the workgroup kernel's 68.  Beyond
  // that a single wave ISSUES too slowly -- measured (tools/bench_soft_nms.py, linear): 100 boxes 80 vs 89 us, 200: 220 vs
  // 154, 500 boxes (8 per lane): 915 vs 350 us: ~2 us per round, every instruction of the round's eight decays waits its
  // turn on one SIMD, where the workgroup kernel spreads them over sixteen waves.  shape 4 forces it up to 512 boxes (tests).
  if ((shape == 0 && n <= kSoftWaveMax) || (shape == 4 && n <= 512)) {
    if (n <= 64) OBB_SOFT_WAVE(1); else if (n <= 128) OBB_SOFT_WAVE(2); else if (n <= 256) OBB_SOFT_WAVE(4); else OBB_SOFT_WAVE(8);
    return true;
  }
#undef OBB_SOFT_WAVE
  if (shape == 1 && n <= 2048) {
    if (n <= 256) OBB_SOFT_FAST(256, 1); else if (n <= 512) OBB_SOFT_FAST(256, 2); else if (n <= 1024) OBB_SOFT_FAST(256, 4); else OBB_SOFT_FAST(256, 8);
  } else if (shape == 2 && n <= kSoftFastMax) {      // (512 x 8 holds kSoftFastMax boxes: larger arrays fall through to `return false`)
    if (n <= 512) OBB_SOFT_FAST(512, 1); else if (n <= 1024) OBB_SOFT_FAST(512, 2); else if (n <= 2048) OBB_SOFT_FAST(512, 4); else OBB_SOFT_FAST(512, 8);
  }
  else if (n <= 256) OBB_SOFT_FAST(256, 1);
  else if (n <= 512) OBB_SOFT_FAST(512, 1);
  else if (n <= 1024) OBB_SOFT_FAST(1024, 1);
  else if (n <= 2048) OBB_SOFT_FAST(1024, 2);
  else if (n <= kSoftFastMax) OBB_SOFT_FAST(1024, 4);
  else return false;
#undef OBB_SOFT_FAST
  return true;
}

constexpr int kSoftRegMaxF32 = 8192, kSoftRegMaxF64 = 4096;
// the register kernel sized by the (largest) array; false = too large for it (float: the global-memory kernel serves it)
template <class T>
static bool soft_launch_reg(const T *dets, int64_t nmax, unsigned groups, float thr, int method, float sigma, float min_score,
                            T *out_dets, int64_t *out_inds, long long *result, const int64_t *seg, hipStream_t st) {
  const int n = seg ? 0 : (int)nmax;
#define OBB_SOFT_REG(TH, NB) hipLaunchKernelGGL((soft_nms_reg_kernel<T, TH, NB>), dim3(groups), dim3(TH), 0, st, dets, n, thr, method, sigma, min_score, out_dets, out_inds, result, seg)
  if (nmax <= 512) OBB_SOFT_REG(256, 2);
  else if (nmax <= 1024) OBB_SOFT_REG(256, 4);
  else if (nmax <= 2048) OBB_SOFT_REG(256, 8);
  else if (nmax <= (sizeof(T) == 4 ? kSoftRegMaxF32 : kSoftRegMaxF64)) {
    if constexpr (sizeof(T) == 4) OBB_SOFT_REG(1024, 8); else OBB_SOFT_REG(1024, 4);
  } else return false;
#undef OBB_SOFT_REG
  return true;
}

}  // namespace obb

extern "C" size_t obb_soft_nms_workspace_bytes(int64_t n) {
  if (n <= obb::kSoftLds) return 256;
  return obb::align_up((size_t)n * 8 * sizeof(float), 256);
}

extern "C" int obb_soft_nms_f32(const float *dets_dev, int64_t n, float iou_thr, int method, float sigma,
                                float min_score, float *out_dets_dev, int64_t *out_inds_dev,
                                int64_t *count_host, void *workspace_dev, size_t workspace_bytes,
                                void *stream) {
  using namespace obb;
  if (count_host) *count_host = 0;
  if (n < 0) return set_err(OBB_EINVAL, "obb_soft_nms_f32: negative n");
  if (method < 0 || method > 2) return set_err(OBB_EINVAL, "obb_soft_nms_f32: bad method %d (0 hard, 1 linear, 2 gaussian)", method);
  if (n == 0) return OBB_OK;
  if (n > (1 << 22)) return set_err(OBB_EINVAL, "obb_soft_nms_f32: n = %lld exceeds %d", (long long)n, 1 << 22);
  if (!dets_dev || !out_dets_dev || !out_inds_dev || !count_host || !workspace_dev)
    return set_err(OBB_EINVAL, "obb_soft_nms_f32: null pointer");
  if (workspace_bytes < obb_soft_nms_workspace_bytes(n))
    return set_err(OBB_EWORKSPACE, "obb_soft_nms_f32: workspace %zu bytes < %zu", workspace_bytes, obb_soft_nms_workspace_bytes(n));
  // K comes back through one pinned, device-mapped word per calling thread
  static thread_local long long *pinned = nullptr;
  if (!pinned) {
    void *p = nullptr;
    OBB_HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped));
    pinned = static_cast<long long *>(p);
  }
  long long *pinned_dev = nullptr;
  OBB_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), pinned, 0));
  hipStream_t st = (hipStream_t)stream;
  float *ws = static_cast<float *>(workspace_dev);
  // (the register-resident kernel below serves float64; for float rows it measured SLOWER than the LDS-array kernel --
  // 6.3 vs 3.3 ms at n = 2000, 21 vs ~12 ms at n = 5000: with 8 boxes per lane a round is ~600 dependent VALU
  // instructions on one wave per SIMD, DESIGN.md 3.4 -- so float keeps the array kernel; OBB_TUNE_SOFT_NMS_REG = 1 forces
  // the register kernel for tests)
  const int64_t knob = tuning(OBB_TUNE_SOFT_NMS_REG, 0);      // 0 default, 1 the position-tracking register kernel, 2 the array kernel only
  // up to 4096 boxes: the kernel without positions first (unique maxima, no NaN: nearly every call); it gives up with
  // result -2 where the reference's array permutation matters, and the call runs again on the array kernel
  if (knob == 0 && soft_launch_fast(dets_dev, n, iou_thr, method, sigma, min_score, out_dets_dev, out_inds_dev, pinned_dev, st)) {
    OBB_LAUNCH_CHECK();
    OBB_HIP_TRY(hipStreamSynchronize(st));
    if (*pinned != kSoftFastGiveUp) { *count_host = *pinned; g_soft_path = 2; return OBB_OK; }
    g_soft_path = 3;
  } else g_soft_path = knob == 1 ? 1 : 0;
  if (knob == 1 && soft_launch_reg<float>(dets_dev, n, 1, iou_thr, method, sigma, min_score, out_dets_dev, out_inds_dev, pinned_dev, nullptr, st)) {
  } else if (n <= 768)          // measured (tools/bench_soft_nms.py): 1.15 us per round with 4 waves, 1.6 with 16 at n = 500
    hipLaunchKernelGGL((soft_nms_kernel<true, 256>), dim3(1), dim3(256), 0, st, dets_dev, (int)n, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, pinned_dev, nullptr);
  else if (n <= kSoftLds)   // 2.0 us per round with 16 waves, 2.3 with 4 at n = 2000
    hipLaunchKernelGGL((soft_nms_kernel<true, 1024>), dim3(1), dim3(1024), 0, st, dets_dev, (int)n, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, pinned_dev, nullptr);
  else
    hipLaunchKernelGGL((soft_nms_kernel<false, 1024>), dim3(1), dim3(1024), 0, st, dets_dev, (int)n, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, pinned_dev, nullptr);
  OBB_LAUNCH_CHECK();
  OBB_HIP_TRY(hipStreamSynchronize(st));
  *count_host = *pinned;
  return OBB_OK;
}

extern "C" int obb_soft_nms_last_path(void) { return obb::g_soft_path; }
#ifdef OBB_SOFT_STAMPS
extern "C" __attribute__((visibility("default"))) int obb_debug_soft_stamps(unsigned long long *host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_soft_stamps), sizeof(unsigned long long) * (size_t)words);
}
#endif

extern "C" int obb_soft_nms_f64(const double *dets_dev, int64_t n, float iou_thr, int method, float sigma,
                                float min_score, double *out_dets_dev, int64_t *out_inds_dev,
                                int64_t *count_host, void *stream) {
  using namespace obb;
  if (count_host) *count_host = 0;
  if (n < 0) return set_err(OBB_EINVAL, "obb_soft_nms_f64: negative n");
  if (method < 0 || method > 2) return set_err(OBB_EINVAL, "obb_soft_nms_f64: bad method %d (0 hard, 1 linear, 2 gaussian)", method);
  if (n == 0) return OBB_OK;
  if (n > kSoftRegMaxF64) return set_err(OBB_EINVAL, "obb_soft_nms_f64: n = %lld exceeds %d (the float64 state lives in registers)", (long long)n, kSoftRegMaxF64);
  if (!dets_dev || !out_dets_dev || !out_inds_dev || !count_host) return set_err(OBB_EINVAL, "obb_soft_nms_f64: null pointer");
  static thread_local long long *pinned = nullptr;
  if (!pinned) {
    void *p = nullptr;
    OBB_HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped));
    pinned = static_cast<long long *>(p);
  }
  long long *pinned_dev = nullptr;
  OBB_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), pinned, 0));
  hipStream_t st = (hipStream_t)stream;
  // up to 2048 rows: the kernel without positions first (as for float rows); it gives up where the array permutation matters
  if (tuning(OBB_TUNE_SOFT_NMS_REG, 0) == 0 &&
      soft_launch_fast64(dets_dev, n, iou_thr, method, sigma, min_score, out_dets_dev, out_inds_dev, pinned_dev, st)) {
    OBB_LAUNCH_CHECK();
    OBB_HIP_TRY(hipStreamSynchronize(st));
    if (*pinned != kSoftFastGiveUp) { *count_host = *pinned; g_soft_path = 2; return OBB_OK; }
    g_soft_path = 3;
  } else g_soft_path = 1;
  soft_launch_reg<double>(dets_dev, n, 1, iou_thr, method, sigma, min_score, out_dets_dev, out_inds_dev, pinned_dev, nullptr, st);
  OBB_LAUNCH_CHECK();
  OBB_HIP_TRY(hipStreamSynchronize(st));
  *count_host = *pinned;
  return OBB_OK;
}

extern "C" int obb_soft_nms_grouped_f32(const float *dets_dev, int64_t n, const int64_t *seg_dev, int64_t num_groups,
                                        float iou_thr, int method, float sigma, float min_score,
                                        float *out_dets_dev, int64_t *out_inds_dev, int64_t *counts_dev,
                                        void *workspace_dev, size_t workspace_bytes, void *stream) {
  return obb_soft_nms_grouped_max_f32(dets_dev, n, seg_dev, num_groups, n, iou_thr, method, sigma, min_score, out_dets_dev,
                                      out_inds_dev, counts_dev, workspace_dev, workspace_bytes, stream);
}

extern "C" int obb_soft_nms_grouped_max_f32(const float *dets_dev, int64_t n, const int64_t *seg_dev, int64_t num_groups,
                                            int64_t max_group, float iou_thr, int method, float sigma, float min_score,
                                            float *out_dets_dev, int64_t *out_inds_dev, int64_t *counts_dev,
                                            void *workspace_dev, size_t workspace_bytes, void *stream) {
  using namespace obb;
  const char *who = "obb_soft_nms_grouped_f32";
  if (max_group < 0 || max_group > n) max_group = n;
  if (n < 0 || num_groups < 0) return set_err(OBB_EINVAL, "%s: negative size", who);
  if (method < 0 || method > 2) return set_err(OBB_EINVAL, "%s: bad method %d (0 hard, 1 linear, 2 gaussian)", who, method);
  if (num_groups == 0) return OBB_OK;
  if (n > (1 << 22) || num_groups > (1 << 20)) return set_err(OBB_EINVAL, "%s: n = %lld / %lld groups too large", who, (long long)n, (long long)num_groups);
  if (!seg_dev || !counts_dev || (n > 0 && (!dets_dev || !out_dets_dev || !out_inds_dev || !workspace_dev)))
    return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (n > 0 && workspace_bytes < obb_soft_nms_workspace_bytes(n))
    return set_err(OBB_EWORKSPACE, "%s: workspace %zu bytes < %zu", who, workspace_bytes, obb_soft_nms_workspace_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  float *ws = static_cast<float *>(workspace_dev);
  long long *res = reinterpret_cast<long long *>(counts_dev);
  const unsigned g = (unsigned)num_groups;
  // `max_group`: the caller's bound on the rows of one group (the segment offsets live on the device): every group's state
  // in LDS while it is <= kSoftLds -- with the total as the only bound, 15 classes of 400 boxes ran on the global-memory
  // state (2.4 us per round instead of 1.2).  A group that exceeds the bound reports -1 in its count.
  // the position-free kernel first, a workgroup per group (a round of 0.7 us instead of 1.2); the groups it gives up on -- two
  // remaining boxes sharing the maximum, a NaN score -- say so in their count and the array kernel's launch behind it runs
  // those only (the others leave at once): no host in between
  int first = 0;
  if (tuning(OBB_TUNE_SOFT_NMS_REG, 0) == 0 && max_group >= 1 &&
      soft_launch_fast(dets_dev, max_group, iou_thr, method, sigma, min_score, out_dets_dev, out_inds_dev, res, st, seg_dev, g)) {
    OBB_LAUNCH_CHECK();
    first = -1;
  }
  if (max_group <= 768)
    hipLaunchKernelGGL((soft_nms_kernel<true, 256>), dim3(g), dim3(256), 0, st, dets_dev, first, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, res, seg_dev);
  else if (max_group <= kSoftLds)
    hipLaunchKernelGGL((soft_nms_kernel<true, 1024>), dim3(g), dim3(1024), 0, st, dets_dev, first, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, res, seg_dev);
  else
    hipLaunchKernelGGL((soft_nms_kernel<false, 1024>), dim3(g), dim3(1024), 0, st, dets_dev, first, iou_thr, method, sigma, min_score, ws, out_dets_dev, out_inds_dev, res, seg_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
