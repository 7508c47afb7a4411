// Optional HIP-event bracket around a kernel class (see abi.cpp).  Host side only.
#pragma once
namespace im360 {
enum ProfKind { PROF_ATTN = 0, PROF_TEMPORAL = 1, PROF_CONV = 2, PROF_GN_STATS = 3, PROF_GN_APPLY = 4, PROF_MISC = 5, PROF_GEMM = 6 };
struct ProfScope {
    ProfScope(int kind, void* stream);
    ~ProfScope();
    int slot_;
    void* stream_;
};
}  // namespace im360
