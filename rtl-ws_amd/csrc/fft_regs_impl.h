// fft_regs_impl.h -- the text of the register-resident FFT building blocks, written once
// for both precisions.  Included by fft_regs.h (float, namespace rtlws -- the f32 fused
// kernels) and fft_regs_f64.h (double, namespace rtlws::f64 -- spectrum_f64_fused.hip) with
//   RTLWS_FR_REAL  float | double          RTLWS_FR_C2    float2 | double2
//   RTLWS_FR_MAKE  make_float2 | make_double2
//   RTLWS_FR_FMA   fmaf | fma              RTLWS_FR_LIT(x)  x##f | x
// No include guard on purpose.  "f2" is the complex type of the instantiation.

typedef RTLWS_FR_REAL real;
typedef RTLWS_FR_C2 f2;

__device__ __forceinline__ f2 mk(real x, real y) { return RTLWS_FR_MAKE(x, y); }
__device__ __forceinline__ f2 cadd(f2 a, f2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ f2 csub(f2 a, f2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ f2 cmul(f2 a, f2 w)
{
    return mk(RTLWS_FR_FMA(-a.y, w.y, a.x * w.x), RTLWS_FR_FMA(a.y, w.x, a.x * w.y));
}

// Forward (sign -1) radix-4 butterfly, natural order in and out.
__device__ __forceinline__ void bfly4(f2& a0, f2& a1, f2& a2, f2& a3)
{
    const f2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = csub(a1, a3);
    a0 = cadd(s0, s2);
    a2 = csub(s0, s2);
    a1 = mk(s1.x + s3.y, s1.y - s3.x);   // s1 - i*s3
    a3 = mk(s1.x - s3.y, s1.y + s3.x);   // s1 + i*s3
}

__device__ __forceinline__ void bfly2(f2& a0, f2& a1)
{
    const f2 s = cadd(a0, a1), d = csub(a0, a1);
    a0 = s;
    a1 = d;
}

// a * exp(-2*pi*i*E/16) for the exponents a 16-point transform needs.
template <int E>
__device__ __forceinline__ f2 mul_w16(f2 a)
{
    constexpr real C1 = RTLWS_FR_LIT(0.92387953251128675613);   // cos(pi/8)
    constexpr real S1 = RTLWS_FR_LIT(0.38268343236508977173);   // sin(pi/8)
    constexpr real H = RTLWS_FR_LIT(0.70710678118654752440);    // sqrt(1/2)
    if constexpr (E == 0) return a;
    else if constexpr (E == 1) return cmul(a, mk(C1, -S1));
    else if constexpr (E == 2) return mk((a.x + a.y) * H, (a.y - a.x) * H);
    else if constexpr (E == 3) return cmul(a, mk(S1, -C1));
    else if constexpr (E == 4) return mk(a.y, -a.x);
    else if constexpr (E == 6) return mk((a.y - a.x) * H, -(a.x + a.y) * H);
    else if constexpr (E == 9) return cmul(a, mk(-C1, S1));
    else { static_assert(E < 0, "unsupported W16 exponent"); return a; }
}

// 16-point forward DFT in registers.  Input slot n holds x[n]; output slot s
// holds X[4*(s&3) + (s>>2)] (digit-reversed), see rev16().
__device__ __forceinline__ void fft16(f2 (&v)[16])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly4(v[m], v[4 + m], v[8 + m], v[12 + m]);
    v[5] = mul_w16<1>(v[5]);
    v[6] = mul_w16<2>(v[6]);
    v[7] = mul_w16<3>(v[7]);
    v[9] = mul_w16<2>(v[9]);
    v[10] = mul_w16<4>(v[10]);
    v[11] = mul_w16<6>(v[11]);
    v[13] = mul_w16<3>(v[13]);
    v[14] = mul_w16<6>(v[14]);
    v[15] = mul_w16<9>(v[15]);
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

__host__ __device__ constexpr int rev16(int s) { return 4 * (s & 3) + (s >> 2); }
__host__ __device__ constexpr int rev8(int s) { return 4 * (s & 1) + (s >> 1); }

// ---- last pass: twiddled radix-R3 in fused-multiply-add form ----------------
// X_p = sum_m (alpha^m z_m) W_R^(m p): the pre-twiddle is a geometric sequence in
// the register index m (alpha = W_T^q2 is the lane's), so the transform is a
// radix-2 decimation-in-time recursion whose butterflies are  a +- w*b  with
// w = alpha^(R/L) * W_L^p at sub-size L.  With w kept as (c, r = s/c) such a
// butterfly is 6 FMAs (t = b*(1 + i r); a +- c*t) instead of a 4-operation
// complex multiply plus 4 additions, and W_L^(p + L/4) = -i W_L^p reuses the
// pair of p: R/2 pairs per lane (1 + 1 + 2 + 4 for L = 2, 4, 8, 16) instead of
// R complex twiddles.  cos = 0 is stored as 1e-20 with r = +-1e20 (the product
// c*t is then exact to 1e-20 relative); the host builds the table.
__device__ __forceinline__ void bfly_tw(f2& a, f2& b, const f2 cr)       // a +- w b
{
    const real tx = RTLWS_FR_FMA(-cr.y, b.y, b.x), ty = RTLWS_FR_FMA(cr.y, b.x, b.y);
    const f2 a0 = a;
    a = mk(RTLWS_FR_FMA(cr.x, tx, a0.x), RTLWS_FR_FMA(cr.x, ty, a0.y));
    b = mk(RTLWS_FR_FMA(-cr.x, tx, a0.x), RTLWS_FR_FMA(-cr.x, ty, a0.y));
}
__device__ __forceinline__ void bfly_tw_rot(f2& a, f2& b, const f2 cr)   // a -+ i w b
{
    const real tx = RTLWS_FR_FMA(-cr.y, b.y, b.x), ty = RTLWS_FR_FMA(cr.y, b.x, b.y);
    const f2 a0 = a;
    a = mk(RTLWS_FR_FMA(cr.x, ty, a0.x), RTLWS_FR_FMA(-cr.x, tx, a0.y));
    b = mk(RTLWS_FR_FMA(-cr.x, ty, a0.x), RTLWS_FR_FMA(cr.x, tx, a0.y));
}

// 16-point forward DFT in registers, second stage in fused-multiply-add form.
// Same input / output convention as fft16() above.  Row q of the second stage is
//   X_p = sum_m W16^(q m) a_m (-i)^(m p),  a_m = first-stage outputs;
// with w1 = W16^q, w2 = W16^(2q) (and W16^(3q) / w1 = w2):
//   s0, s1 = a0 +- w2 a2        u, u' = a1 +- w2 a3
//   X0, X2 = s0 +- w1 u         X1, X3 = s1 -+ i w1 u'
// -- four "a +- w b" butterflies of 6 FMAs each with w as (cos, tan) literals,
// 24 operations per row against 12 (three complex multiplies) + 16 for "twiddle,
// then butterfly"; row 2 has w2 = -i (20 operations).  148 operations per
// transform instead of 160.
// (fft16_fma_rows is that second stage alone: for callers that produce the first stage's outputs themselves)
__device__ __forceinline__ void fft16_fma_rows(f2 (&v)[16])
{
    constexpr real C1 = RTLWS_FR_LIT(0.92387953251128675613);    // cos(pi/8)
    constexpr real T1 = -RTLWS_FR_LIT(0.41421356237309504880);   // tan(-pi/8)
    constexpr real H = RTLWS_FR_LIT(0.70710678118654752440);     // cos(pi/4)
    constexpr real C3 = RTLWS_FR_LIT(0.38268343236508977173);    // cos(3 pi/8)
    constexpr real T3 = -RTLWS_FR_LIT(2.41421356237309504880);   // tan(-3 pi/8)
    bfly4(v[0], v[1], v[2], v[3]);                                   // q = 0
    // q = 1: w1 = W16^1 = (C1, T1), w2 = W16^2 = (H, -1)
    bfly_tw(v[4], v[6], mk(H, -RTLWS_FR_LIT(1.0)));
    bfly_tw(v[5], v[7], mk(H, -RTLWS_FR_LIT(1.0)));
    bfly_tw(v[4], v[5], mk(C1, T1));                                 // X0 -> v[4], X2 -> v[5]
    bfly_tw_rot(v[6], v[7], mk(C1, T1));                             // X1 -> v[6], X3 -> v[7]
    { const f2 x2 = v[5]; v[5] = v[6]; v[6] = x2; }                  // natural order X0..X3
    // q = 2: w1 = W16^2 = (H, -1), w2 = W16^4 = -i
    {
        const f2 a0 = v[8], a1 = v[9], a2 = v[10], a3 = v[11];
        const f2 s0 = mk(a0.x + a2.y, a0.y - a2.x), s1 = mk(a0.x - a2.y, a0.y + a2.x);   // a0 -+ i a2
        const f2 u = mk(a1.x + a3.y, a1.y - a3.x), up = mk(a1.x - a3.y, a1.y + a3.x);    // a1 -+ i a3
        f2 p0 = s0, p1 = u, q0 = s1, q1 = up;
        bfly_tw(p0, p1, mk(H, -RTLWS_FR_LIT(1.0)));
        bfly_tw_rot(q0, q1, mk(H, -RTLWS_FR_LIT(1.0)));
        v[8] = p0; v[9] = q0; v[10] = p1; v[11] = q1;
    }
    // q = 3: w1 = W16^3 = (C3, T3), w2 = W16^6 = (-H, +1)
    bfly_tw(v[12], v[14], mk(-H, RTLWS_FR_LIT(1.0)));
    bfly_tw(v[13], v[15], mk(-H, RTLWS_FR_LIT(1.0)));
    bfly_tw(v[12], v[13], mk(C3, T3));
    bfly_tw_rot(v[14], v[15], mk(C3, T3));
    { const f2 x2 = v[13]; v[13] = v[14]; v[14] = x2; }
}

__device__ __forceinline__ void fft16_fma(f2 (&v)[16])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly4(v[m], v[4 + m], v[8 + m], v[12 + m]);
    fft16_fma_rows(v);
}

// (the multiply-then-butterfly form fft16() is the A/B partner: tools/variants/csrc_hooks.patch, -DRTLWS_FFT16_FMA=0)
__device__ __forceinline__ void fft16_sel(f2 (&v)[16])
{
    fft16_fma(v);
}

template <int R>
__host__ __device__ constexpr int bitrev(int i)
{
    int o = 0;
    for (int b = 1; b < R; b <<= 1) { o = (o << 1) | (i & 1); i >>= 1; }
    return o;
}
// first pair of sub-size L in a lane's table: L = 2 -> 0, 4 -> 1, 8 -> 2, 16 -> 4
__host__ __device__ constexpr int tw_pair_base(int L) { return L == 2 ? 0 : L / 4; }

template <int R3>
__host__ __device__ constexpr int rev_last(int s);

// Radix-R3 over the R3 consecutive registers starting at v[base] (slot m holds
// z_m); output slot s holds X[rev_last<R3>(s)].
template <int R3>
__device__ __forceinline__ void fft_last(f2 (&v)[16], int base, const f2 (&tw)[R3 / 2])
{
    f2 y[R3];
#pragma unroll
    for (int i = 0; i < R3; ++i) y[i] = v[base + bitrev<R3>(i)];
#pragma unroll
    for (int L = 2; L <= R3; L *= 2) {
        const int half = L / 2, quarter = L >= 4 ? L / 4 : 1;
#pragma unroll
        for (int b = 0; b < R3; b += L)
#pragma unroll
            for (int p = 0; p < half; ++p) {
                if (p < quarter) bfly_tw(y[b + p], y[b + p + half], tw[tw_pair_base(L) + p]);
                else bfly_tw_rot(y[b + p], y[b + p + half], tw[tw_pair_base(L) + p - quarter]);
            }
    }
#pragma unroll
    for (int s = 0; s < R3; ++s) v[base + s] = y[rev_last<R3>(s)];
}

template <int R3>
__host__ __device__ constexpr int rev_last(int s)
{
    return R3 == 4 ? s : (R3 == 8 ? rev8(s) : rev16(s));
}

// ---- Hann window from two lane constants -----------------------------------------
// Thread t holds x[T*r + t], r < 16, and N = 16*T, so the periodic Hann weight is
//   w_r = 0.5 - 0.5*cos(2*pi*r/16 + theta_t) = 0.5 - cos16(r)*ch + sin16(r)*sh,
// ch = 0.5*cos(theta_t), sh = 0.5*sin(theta_t), theta_t = 2*pi*t/N: two lane
// constants (host table, f64-computed) and two FMAs per weight, evaluated once per
// persistent workgroup into sixteen registers (no N-entry table, no sixteen
// strided loads).  Absolute error of a weight <= 1.2e-7 (1 ulp of 0.5 plus the
// table rounding), i.e. <= 2e-7 of a sample, orders below the f32 transform's own
// error ("Error budget").
__host__ __device__ constexpr real cos16(int r)
{
    constexpr real C1 = RTLWS_FR_LIT(0.92387953251128675613), S1 = RTLWS_FR_LIT(0.38268343236508977173), H = RTLWS_FR_LIT(0.70710678118654752440);
    switch (r & 15) {
    case 0: return RTLWS_FR_LIT(1.0);   case 1: return C1;   case 2: return H;    case 3: return S1;
    case 4: return RTLWS_FR_LIT(0.0);   case 5: return -S1;  case 6: return -H;   case 7: return -C1;
    case 8: return -RTLWS_FR_LIT(1.0);  case 9: return -C1;  case 10: return -H;  case 11: return -S1;
    case 12: return RTLWS_FR_LIT(0.0);  case 13: return S1;  case 14: return H;   default: return C1;
    }
}
__host__ __device__ constexpr real sin16(int r) { return cos16(r + 12); }   // sin(x) = cos(x - pi/2)

__device__ __forceinline__ real hann_w(int r, const f2 wcs)
{
    return RTLWS_FR_FMA(-cos16(r), wcs.x, RTLWS_FR_FMA(sin16(r), wcs.y, RTLWS_FR_LIT(0.5)));
}

// The same window INSIDE the first butterfly layer of a lane's radix-16.  That layer combines the slots
// m, 4+m, 8+m, 12+m, whose weights are a quarter turn apart: with phi = theta_t + 2 pi m / 16
//   w = (1 - cos phi, 1 + sin phi, 1 + cos phi, 1 - sin phi) / 2
// so the layer's first half on the weighted points is, from the UNWEIGHTED sums and differences
// e0 = x0 + x2, e1 = x0 - x2, o0 = x1 + x3, o1 = x1 - x3 (integers for integer samples),
//   2 (w0 x0 + w2 x2) = e0 - cos phi e1      2 (w0 x0 - w2 x2) = e1 - cos phi e0
//   2 (w1 x1 + w3 x3) = o0 + sin phi o1      2 (w1 x1 - w3 x3) = o1 + sin phi o0
// -- one FMA per component where the unwindowed layer has one addition: the window costs nothing but
// (cos phi, sin phi) for m < 4, two operations each from the lane's (0.5 cos, 0.5 sin) theta_t.  The factor 2
// is the caller's (it rides on the pass-1 twiddles).
__device__ __forceinline__ f2 hann_cs_m(int m, const f2 wcs)         // (cos, sin)(theta_t + 2 pi m / 16)
{
    const real c2 = RTLWS_FR_LIT(2.0) * cos16(m), s2 = RTLWS_FR_LIT(2.0) * sin16(m);
    return mk(RTLWS_FR_FMA(-s2, wcs.y, c2 * wcs.x), RTLWS_FR_FMA(s2, wcs.x, c2 * wcs.y));
}
__device__ __forceinline__ void hann_bfly4(f2 e0, f2 e1, f2 o0, f2 o1, const f2 cs, f2& a0, f2& a1, f2& a2, f2& a3)
{
    const f2 s0 = mk(RTLWS_FR_FMA(-cs.x, e1.x, e0.x), RTLWS_FR_FMA(-cs.x, e1.y, e0.y));
    const f2 s1 = mk(RTLWS_FR_FMA(-cs.x, e0.x, e1.x), RTLWS_FR_FMA(-cs.x, e0.y, e1.y));
    const f2 s2 = mk(RTLWS_FR_FMA(cs.y, o1.x, o0.x), RTLWS_FR_FMA(cs.y, o1.y, o0.y));
    const f2 s3 = mk(RTLWS_FR_FMA(cs.y, o0.x, o1.x), RTLWS_FR_FMA(cs.y, o0.y, o1.y));
    a0 = cadd(s0, s2);
    a2 = csub(s0, s2);
    a1 = mk(s1.x + s3.y, s1.y - s3.x);   // s1 - i*s3
    a3 = mk(s1.x - s3.y, s1.y + s3.x);   // s1 + i*s3
}

