// Fr stage of the prover: QAP.eval (src/lib/zk/QAP.ml:120-135) for sparse R1CS on the GPU.
#pragma once
#include "rns_ntt.cuh"
#include "zk_common.h"

#include <vector>

namespace zk {

struct CsrDev {
    DevBuf ptr, col, val;   // val in Montgomery form
    uint64_t nnz = 0;
};

struct FrStage {
    uint32_t n = 0, m = 0;
    uint32_t n2 = 0, log_n2 = 0;     // next power of two >= n : size of the basis-conversion tree
    uint32_t S = 0, log_S = 0;       // 2 * n2 : NTT size of the convolutions
    CsrDev L, R, O;
    DevBuf invfact;                   // 1/i!, i < n2
    DevBuf e_ntt;                     // NTT_S of (-1)^j / j!, bit-reversed order
    DevBuf pntt;                      // log_n2 levels x n2: NTT_len(P_{s,len/2}) / len per node
    DevBuf iz_ntt;                    // NTT_S of (rev Z)^-1 mod x^(n-1)
    DevBuf z;                         // Z coefficients, n + 1 (Montgomery)
    // Lagrange-form keys only (frstage_init_lagrange): h through its VALUES on the shifted points n..2n-2
    bool lagrange = false;
    DevBuf g_ntt;                     // NTT_S of g[e] = 1/e (g[0] = 0): kernel of the extrapolation convolution
    DevBuf zt;                        // Z(n + t) = (n+t)!/t!, t < n - 1
    // round 4: the same fixed factors for the convolutions through the residue number system (rns_ntt.cuh), built when the transforms fit it (log_S <= 23)
    bool rns_ok = false;
    DevBuf e_rns, iz_rns, g_rns;      // counterparts of e_ntt, iz_ntt, g_ntt: 18 x S residues
    std::vector<DevBuf> p_rns;        // p_rns[l]: the level-l table of the subproduct tree (nodes of 2^l points), 18 x n2 residues, for the levels above the fused ones
    uint32_t rns_first_level = 0;     // lowest level with an RNS table
};
// per-proof scratch: one per proof in flight
struct FrScratch {
    DevBuf wit, abc, d, tmp, bufA, h, flag;      // bufA: two convolution buffers of S elements back to back
    RnsWork rns_a, rns_b;                        // residue arrays of the convolutions (allocated at first use)
};

int frstage_init(FrStage& f, uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, hipStream_t s);
int frstage_scratch_alloc(const FrStage& f, FrScratch& sc);
// witness: m canonical Fr on device.  Leaves v = sc.d[0..n), w = sc.d[n2..n2+n), h = sc.h[0..n-1) in
// Montgomery form; *sc.flag |= 1 when some gate is violated (QAP.ml:134), |= 2 when a witness
// value is not canonical.  Everything is enqueued on `s`; nothing synchronizes.
int frstage_eval(const FrStage& f, FrScratch& sc, const void* d_witness_canonical, hipStream_t s);

// Lagrange-form variant (scope row f4): no basis conversion at all.  Leaves the VALUES a = L w, b = R w in
// sc.abc[0..n), sc.abc[n..2n) and h(n + t), t < n - 1, in sc.h (Montgomery); flags as above.
int frstage_init_lagrange(FrStage& f, hipStream_t s);     // after frstage_init
int frstage_eval_lagrange(const FrStage& f, FrScratch& sc, const void* d_witness_canonical, hipStream_t s);

// kappa[0], kappa[1] (Montgomery, 2 x 32 B on device) = the X^(n-1) coefficients of the interpolants of the VALUES a = sc.abc[0..n), b = sc.abc[n..2n):
// sum_i x_i c_i with the barycentric weights c_i = (-1)^(n-1-i) / (i! (n-1-i)!) -- after either frstage_eval form.  d_partials: 2 * LEAD_BLOCKS Fr of
// scratch.  (Pinocchio's compact h pool, pinocchio.hip: v = kappa X^(n-1) + a polynomial of degree <= n-2.)
static constexpr uint32_t LEAD_BLOCKS = 128;
int frstage_leading_coeffs(const FrStage& f, const FrScratch& sc, void* d_kappa, void* d_partials, hipStream_t s);
// out[t] = (n + t)^e for t < cnt (Montgomery)
int frstage_shifted_powers(void* d_out, uint32_t n, uint32_t e, uint32_t cnt, hipStream_t s);

// subproduct-tree tables over the points offset .. offset + n2 - 1 (plain Montgomery form, see frstage.hip)
int frstage_tree_tables(uint32_t n2, uint32_t log_n2, uint32_t offset, void* pntt, DevBuf& q, hipStream_t s, std::vector<DevBuf>* p_rns = nullptr, uint32_t rns_first_level = 0);

// a*b via NTT on device (Montgomery in/out); out must hold na+nb-1 elements
int dev_poly_mul(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, void* d_out, hipStream_t s);

}  // namespace zk
