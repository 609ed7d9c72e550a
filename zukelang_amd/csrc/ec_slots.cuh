// The XYZZ group law with ONE POINT SPREAD OVER FOUR SLOTS of a wave (a slot = one lane in G1, a lane pair in G2 -- the Fp2H split of
// ff.cuh): slot s keeps coordinate s of the point (0 = x, 1 = y, 2 = zz, 3 = zzz) and every step of the formula runs four of its field
// products at once, one per slot.
//
// Why: the bucket reduction (msm_tail.hip) is a chain of ~45 DEPENDENT additions executed by a handful of waves.  A wave issues its
// instructions in order whatever the other lanes do, so a lane that multiplies 14 times in a row (add-2008-s on one lane) keeps the whole
// chain waiting 14 product times per link; the 14 products have only four levels of dependency:
//     U1 = X1 ZZ2   S1 = Y1 ZZZ2   U2 = ZZ1 X2   S2 = ZZZ1 Y2          (slot s: own coordinate x coordinate s^2 of q)
//     PP = P^2      RR = R^2       A = ZZ1 ZZ2   B = ZZZ1 ZZZ2         (P = U2 - U1, R = S2 - S1)
//     PPP = P PP    Q = U1 PP      ZZ3 = A PP
//     S1 PPP        R (Q - X3)                   ZZZ3 = B PPP          (X3 = RR - PPP - 2 Q,  Y3 = R (Q - X3) - S1 PPP)
// Four product times per link instead of fourteen, for five operand exchanges between the slots (ds_bpermute, 14 limbs each: ~3 % of the
// instructions of one product step).  Doubling (dbl-2008-s-1) has three levels.  Same formulas, same special cases and therefore the same
// group elements as xyzz_add_impl / xyzz_dbl_impl of ec.cuh; coordinates differ from theirs only by lazy-reduction representatives.
// (The reference's G.add / G.double behind curve.ml:159-191, delegated to opam bls12-381.)  A slot addition executes 16 slot-products and four copies of
// the non-multiply work for 14 useful products: it pays where a launch is bound by its chain, not where it is bound by the chip's throughput
// (msm_tail.cuh says which steps of the reduction are which).
//
// All lanes of a group (4 slots) must be active and follow the same control flow through these functions: every predicate below is made
// group-uniform by fetching it from the slot that owns it.
#pragma once
#include "ec.cuh"

namespace zk {

template <class T> struct SlotGeom;
template <> struct SlotGeom<Fp> { static constexpr uint32_t LP = 1, G = 4; };      // lanes per slot, lanes per point
template <> struct SlotGeom<Fp2H> { static constexpr uint32_t LP = 2, G = 8; };
// the same field with another value bound (ff.cuh: the bound is part of the type, a looser one converts implicitly)
template <class T, int B> struct WithBound;
template <int A, int B> struct WithBound<FpB<A>, B> { using type = FpB<B>; };
template <int A, int B> struct WithBound<Fp2HB<A>, B> { using type = Fp2HB<B>; };
template <int B> FF_INLINE uint32_t* slot_limbs(FpB<B>& a) { return a.v; }
template <int B> FF_INLINE uint32_t* slot_limbs(Fp2HB<B>& a) { return a.v.v; }
template <int B> FF_INLINE const uint32_t* slot_limbs(const FpB<B>& a) { return a.v; }
template <int B> FF_INLINE const uint32_t* slot_limbs(const Fp2HB<B>& a) { return a.v.v; }
template <class T> FF_INLINE uint32_t slot_id() { return (threadIdx.x / SlotGeom<T>::LP) & 3u; }
// ds_bpermute address of the lane that holds the same Fp2 component in slot `src` of this lane's group (workgroups are 1-D multiples of 64)
template <class T> FF_INLINE int slot_addr(uint32_t src) {
    constexpr uint32_t LP = SlotGeom<T>::LP, G = SlotGeom<T>::G;
    const uint32_t lane = threadIdx.x & 63u;
    return (int)(((lane & ~(G - 1)) | (src * LP) | (lane & (LP - 1))) << 2);
}
template <class V> FF_INLINE V slot_fetch(const V& v, int addr) {
    V r;
#pragma unroll
    for (int l = 0; l < FPL; l++) slot_limbs(r)[l] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)slot_limbs(v)[l]);
    return r;
}
FF_INLINE int slot_flag(int f, int addr) { return __builtin_amdgcn_ds_bpermute(addr, f); }
// take_b ? b : a, both widened to the bound of R
template <class R, class A, class B> FF_INLINE R slot_select(bool take_b, const A& a, const B& b) {
    const R ra(a), rb(b);
    R r;
#pragma unroll
    for (int l = 0; l < FPL; l++) slot_limbs(r)[l] = take_b ? slot_limbs(rb)[l] : slot_limbs(ra)[l];
    return r;
}
template <class T> FF_INLINE T slot_zero() { return T(FieldOps<T>::zero()); }

// 2 * (point held as `own` per slot); identity and y = 0 give the identity
template <class T> FF_INLINE T xyzz_dbl_slots(const T& own) {
    using B6 = typename WithBound<T, 6>::type;
    using B10 = typename WithBound<T, 10>::type;
    using B128 = typename WithBound<T, 2 * FP_REST>::type;
    const uint32_t s = slot_id<T>();
    const int a0 = slot_addr<T>(0), a1 = slot_addr<T>(1), a2 = slot_addr<T>(2), a3 = slot_addr<T>(3), ax1 = slot_addr<T>(s ^ 1);
    const int z = fe_is_zero(own) ? 1 : 0;
    if (slot_flag(z, a2) | slot_flag(z, a1)) return slot_zero<T>();
    const auto U = fe_dbl(own);                                              // slot 1: 2 Y
    const B128 f1 = slot_select<B128>(s == 1, own, U);
    const auto m1 = fe_mul(f1, f1);                                          // slot 0: X^2   slot 1: V = U^2
    const auto V = slot_fetch(m1, a1);
    const T X = slot_fetch(own, a0);
    const B6 M = fe_add(fe_dbl(m1), m1);                                     // slot 0: 3 X^2
    const B128 f2a = slot_select<B128>(s == 3, slot_select<B128>(s == 1, slot_select<T>(s == 0, own, M), U), X);
    const B6 f2b = slot_select<B6>(s == 0, V, M);
    const auto m2 = fe_mul(f2a, f2b);                                        // slot 0: M^2   1: W = U V   2: ZZ3 = ZZ V   3: S = X V
    const auto W = slot_fetch(m2, a1), S = slot_fetch(m2, a3);
    const B10 X3 = fe_sub(m2, fe_dbl(S));                                    // slot 0: M^2 - 2 S
    const B6 f3a = slot_select<B6>(s == 0, slot_select<B6>(s == 1, W, m2), M);
    const T f3b = slot_select<T>(s == 0, own, fe_sub(S, X3));
    const auto m3 = fe_mul(f3a, f3b);                                        // slot 0: M (S - X3)   1: W Y   3: ZZZ3 = W ZZZ
    const auto t = slot_fetch(m3, ax1);
    const B6 Y3 = fe_sub(t, m3);                                             // slot 1: M (S - X3) - W Y
    return slot_select<T>(s == 3, slot_select<T>(s == 2, slot_select<T>(s == 1, X3, Y3), m2), m3);
}

// own += q for points spread over the slots.  qs = coordinate s of q (what this slot would keep of it), qx = coordinate s^2 of q.
template <class T> FF_INLINE void xyzz_add_slots(T& own, const T& qs, const T& qx) {
    using B6 = typename WithBound<T, 6>::type;
    using B10 = typename WithBound<T, 10>::type;
    using B18 = typename WithBound<T, 18>::type;
    const uint32_t s = slot_id<T>();
    const int a0 = slot_addr<T>(0), a1 = slot_addr<T>(1), a2 = slot_addr<T>(2), ax1 = slot_addr<T>(s ^ 1), ax2 = slot_addr<T>(s ^ 2);
    const bool low = s < 2;
    {
        const int z = fe_is_zero(slot_select<T>(s == 0, own, qx)) ? 1 : 0;  // slot 0: ZZ2 = 0 ?   slot 2: ZZ1 = 0 ?
        if (slot_flag(z, a0)) return;                                        // q is the identity
        if (slot_flag(z, a2)) {                                              // acc is
            own = qs;
            return;
        }
    }
    const auto m1 = fe_mul(own, qx);                                         // U1 | S1 | U2 | S2
    const auto x1 = slot_fetch(m1, ax2);
    using P2 = std::remove_const_t<decltype(x1)>;
    const P2 lo = slot_select<P2>(low, x1, m1), hi = slot_select<P2>(low, m1, x1);  // slots 0, 2: U1, U2   slots 1, 3: S1, S2
    const B6 d = fe_sub(hi, lo);                                             // slots 0, 2: P        slots 1, 3: R
    {
        const int z = fe_is_zero(d) ? 1 : 0;
        if (slot_flag(z, a0)) {                                              // equal x: P + P or P + (-P)
            if (slot_flag(z, a1)) own = xyzz_dbl_slots(own);
            else own = slot_zero<T>();
            return;
        }
    }
    const auto m2 = fe_mul(slot_select<T>(low, own, d), slot_select<T>(low, qs, d));   // PP | RR | A = ZZ1 ZZ2 | B = ZZZ1 ZZZ2
    const auto PP = slot_fetch(m2, a0);
    const auto u = slot_fetch(lo, ax1);                                      // slot 0: S1   slot 1: U1
    const auto m3 = fe_mul(slot_select<B6>(s == 0, slot_select<B6>(s == 1, m2, u), d), PP);   // PPP | Q = U1 PP | ZZ3 = A PP | (unused)
    const auto PPP = slot_fetch(m3, a0);
    const B10 X3 = fe_sub_sub_dbl(m2, PPP, m3);                              // slot 1: RR - PPP - 2 Q
    const auto m4 = fe_mul(slot_select<B6>(s == 0, slot_select<B6>(s == 1, m2, d), u),
                           slot_select<B18>(s == 1, PPP, fe_sub(m3, X3)));   // S1 PPP | R (Q - X3) | (unused) | ZZZ3 = B PPP
    const B10 w = slot_fetch(slot_select<B10>(s == 0, X3, m4), ax1);         // slot 0 receives X3, slot 1 receives S1 PPP
    const B18 Y3 = fe_sub(m4, w);
    own = slot_select<T>(s == 3, slot_select<T>(s == 2, slot_select<T>(s == 1, w, Y3), m3), m4);
}
}  // namespace zk
