// Optional HIP-event bracket around a kernel class (see abi.cpp).  Host side only.
#pragma once
namespace im360 {
enum ProfKind { PROF_ATTN = 0, PROF_TEMPORAL = 1, PROF_CONV = 2, PROF_GN_STATS = 3, PROF_GN_APPLY = 4, PROF_MISC = 5, PROF_GEMM = 6, PROF_ATTN_WARP = 7, PROF_ATTN_X2 = 8 };      // 7 / 8: the biased (WarpAttn) and the two-set (text + IP) launches, split out of 0
struct ProfScope {
    ProfScope(int kind, void* stream);
    ~ProfScope();
    int slot_;
    void* stream_;
};
}  // namespace im360

// Tuning knobs for A/B measurements (tools/bench_kernels.py); none of them changes results.  Defaults are the measured
// best; the IM360_* environment variables of the same names seed them once at load time, im360_tuning_set() changes them
// at run time.  Plain ints read at launch time by the host-side launchers only.
namespace im360 {
enum Knob { KNOB_ATTN_QB = 0, KNOB_CONV_BIG = 1, KNOB_CONV_BK = 2, KNOB_TATTN_SCALAR = 3, KNOB_CONV_RING = 4, KNOB_ATTN_HL = 5, KNOB_CONV_DBG = 6, KNOB_CONV_HALO = 7, KNOB_CONV_CM = 8, KNOB_LN_PACKED = 9, KNOB_RING_GROUPS = 10, KNOB_ATTN_X = 11, KNOB_ATTN_DS = 12, KNOB_ATTN_ONE = 13, KNOB_ATTN_DBG = 14, KNOB_ATTN_HG = 15, KNOB_CONV_SMALL = 16, KNOB_ATTN_W3 = 17, KNOB_ATTN_PIPE = 18, KNOB_CONV_STAG = 19, KNOB_CONV_PERSIST = 20, KNOB_GN_APPLY = 21, KNOB_TATTN_NT = 22, KNOB_NT = 23, KNOB_G4 = 24, KNOB_GN_WGS = 25, KNOB_CONV_KSPLIT = 26, KNOB_COUNT = 27 };
int knob(int k);
}  // namespace im360
