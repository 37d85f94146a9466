// ronkathon.hpp -- C++ host mirror of ronkathon's field / polynomial surface over the C ABI.
//
// The reference is Rust (nightly) and no Rust toolchain exists in this image, so the host side
// above include/ronk_ntt.h is written in C++ with the SAME names, argument meaning and error
// behaviour as the reference, so that tests read like the reference's own:
//   PrimeField<P>            src/algebra/field/prime/mod.rs:39-140, prime/arithmetic.rs:3-71
//   Field / FiniteField      src/algebra/field/mod.rs:17-76   (ZERO, ONE, inverse, pow, PRIMITIVE_ELEMENT,
//                                                              primitive_root_of_unity)
//   Polynomial<B, F, D>      src/polynomial/mod.rs:34-515, src/polynomial/arithmetic.rs:16-146
// A reference panic becomes a thrown ronkathon::Panic carrying the same message.
// Scalars are host values (like the Rust value type); every array operation goes to the GPU
// through libronk_ntt.so -- there is no CPU implementation behind this header.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ronk_ntt.h"

namespace ronkathon {

struct Panic : std::runtime_error {
  int code;
  explicit Panic(int c) : std::runtime_error(ronk_strerror(c)), code(c) {}
};
inline void check(int rc) { if (rc != RONK_OK) throw Panic(rc); }

// ---- PrimeField<P>: #[repr(transparent)]-like wrapper of the canonical residue
template <uint64_t P>
struct PrimeField {
  uint64_t value = 0;
  static constexpr uint64_t ORDER = P;                       // Finite::ORDER, src/algebra/mod.rs:8-13
  constexpr PrimeField() = default;
  static PrimeField new_(uint64_t v) {                       // prime/mod.rs:48-51 (asserts primality)
    check(ronk_check_prime(P));
    PrimeField f; f.value = v % P; return f;
  }
  PrimeField(uint64_t v) { *this = new_(v); }                // From<usize>
  static PrimeField ZERO() { PrimeField f; f.value = 0; return f; }
  static PrimeField ONE() { PrimeField f; f.value = 1 % P; return f; }
  static PrimeField PRIMITIVE_ELEMENT() {                    // prime/mod.rs:87-90
    uint64_t g; check(ronk_primitive_element(P, &g)); PrimeField f; f.value = g; return f;
  }
  static PrimeField primitive_root_of_unity(uint64_t n) {    // field/mod.rs:70-75
    uint64_t w; check(ronk_root_of_unity(P, PRIMITIVE_ELEMENT().value, n, &w)); PrimeField f; f.value = w; return f;
  }
  PrimeField pow(uint64_t e) const {                         // prime/mod.rs:74-84
    unsigned __int128 r = 1 % P, b = value;
    while (e) { if (e & 1) r = r * b % P; b = b * b % P; e >>= 1; }
    PrimeField f; f.value = (uint64_t)r; return f;
  }
  std::optional<PrimeField> inverse() const {                // prime/mod.rs:62-72
    if (value == 0) return std::nullopt;
    return pow(P - 2);
  }
  friend PrimeField operator+(PrimeField a, PrimeField b) { PrimeField f; f.value = (uint64_t)(((unsigned __int128)a.value + b.value) % P); return f; }
  friend PrimeField operator-(PrimeField a, PrimeField b) { PrimeField f; f.value = a.value >= b.value ? a.value - b.value : a.value - b.value + P; return f; }
  friend PrimeField operator*(PrimeField a, PrimeField b) { PrimeField f; f.value = (uint64_t)((unsigned __int128)a.value * b.value % P); return f; }
  friend PrimeField operator/(PrimeField a, PrimeField b) {  // self * rhs.inverse().unwrap()
    auto i = b.inverse(); if (!i) throw Panic(RONK_ERR_ZERO_INVERSE); return a * *i;
  }
  PrimeField operator-() const { return ZERO() - *this; }
  // FieldExt (field/mod.rs:79-84; prime/mod.rs:142-226)
  bool euler_criterion() const { return pow((P - 1) / 2).value == 1; }
  // Tonelli-Shanks as the reference writes it: (smaller root, larger root); ZERO -> (0, 0); a non-residue is its assert
  std::pair<PrimeField, PrimeField> sqrt() const {
    if (value == 0) return {ZERO(), ZERO()};
    if (P == 2) throw Panic(RONK_ERR_UNSUPPORTED);           // the reference's search for a non-residue never ends over F_2
    if (!euler_criterion()) throw Panic(RONK_ERR_NOT_RESIDUE);
    uint64_t q = P - 1, s = 0;
    while ((q & 1) == 0) { q >>= 1; s++; }
    PrimeField z = new_(2);
    while (z.euler_criterion()) z = z + ONE();
    uint64_t m = s;
    PrimeField c = z.pow(q), t = pow(q), r = pow((q + 1) / 2);
    for (;;) {
      if (t.value == 1) { PrimeField nr = -r; return nr.value < r.value ? std::make_pair(nr, r) : std::make_pair(r, nr); }
      uint64_t i = 1;
      PrimeField t_pow = t.pow(2);
      while (t_pow.value != 1) { t_pow = t_pow.pow(2); i++; }
      PrimeField b = c.pow((uint64_t)1 << (m - i - 1));
      m = i; c = b.pow(2); t = t * c; r = r * b;
    }
  }
  // the same over arrays, on the GPU (ronk_vec_euler / ronk_vec_sqrt)
  static std::vector<uint64_t> vec_euler(const std::vector<PrimeField>& a) {
    std::vector<uint64_t> out(a.size());
    check(ronk_vec_euler(P, reinterpret_cast<const uint64_t*>(a.data()), out.data(), a.size()));
    return out;
  }
  static std::pair<std::vector<PrimeField>, std::vector<PrimeField>> vec_sqrt(const std::vector<PrimeField>& a) {
    std::vector<PrimeField> r0(a.size()), r1(a.size());
    check(ronk_vec_sqrt(P, reinterpret_cast<const uint64_t*>(a.data()), reinterpret_cast<uint64_t*>(r0.data()),
                        reinterpret_cast<uint64_t*>(r1.data()), a.size()));
    return {r0, r1};
  }
  friend bool operator==(PrimeField a, PrimeField b) { return a.value == b.value; }
  friend bool operator!=(PrimeField a, PrimeField b) { return a.value != b.value; }
};
static_assert(sizeof(PrimeField<101>) == sizeof(uint64_t), "PrimeField must be layout-compatible with u64");

using PlutoBaseField = PrimeField<101>;                       // prime/mod.rs:26
using PlutoScalarField = PrimeField<17>;                      // prime/mod.rs:30
using GoldilocksField = PrimeField<RONK_GOLDILOCKS_P>;        // the new 64-bit FiniteField implementor

// ---- bases (polynomial/mod.rs:48-72)
struct Monomial { friend bool operator==(Monomial, Monomial) { return true; } };
template <class F> struct Lagrange {
  std::vector<F> nodes;
  friend bool operator==(const Lagrange& a, const Lagrange& b) { return a.nodes == b.nodes; }
};

template <class B, class F, size_t D> struct Polynomial;

template <class F, size_t D>
struct Polynomial<Lagrange<F>, F, D> {
  std::array<F, D> coefficients;
  Lagrange<F> basis;
  // Polynomial::<Lagrange<F>,F,D>::new (mod.rs:358-365): assert_eq!((ORDER - 1) % n, 0); nodes = [w^i]
  static Polynomial new_(const std::array<F, D>& c) {
    Polynomial p; p.coefficients = c; p.basis.nodes.resize(D);
    check(ronk_lagrange_nodes(F::ORDER, F::PRIMITIVE_ELEMENT().value, reinterpret_cast<uint64_t*>(p.basis.nodes.data()), D));
    return p;
  }
  size_t num_terms() const { return D; }
  F evaluate(F x) const {                                    // mod.rs:382-415 (barycentric)
    F r; check(ronk_lagrange_eval(F::ORDER, reinterpret_cast<const uint64_t*>(coefficients.data()),
                                  reinterpret_cast<const uint64_t*>(basis.nodes.data()), D, x.value, &r.value));
    return r;
  }
  Polynomial<Monomial, F, D> ifft() const;                   // mod.rs:430-453
  friend bool operator==(const Polynomial& a, const Polynomial& b) { return a.coefficients == b.coefficients && a.basis == b.basis; }
};

template <class F, size_t D>
struct Polynomial<Monomial, F, D> {
  std::array<F, D> coefficients;
  Monomial basis;
  static Polynomial new_(const std::array<F, D>& c) { Polynomial p; p.coefficients = c; return p; }   // mod.rs:98
  template <size_t N> static Polynomial from(const std::array<F, N>& c) {                              // mod.rs:503-515
    Polynomial p; p.coefficients.fill(F::ZERO());
    for (size_t i = 0; i < (N < D ? N : D); i++) p.coefficients[i] = c[i];
    return p;
  }
  const uint64_t* raw() const { return reinterpret_cast<const uint64_t*>(coefficients.data()); }
  uint64_t* raw() { return reinterpret_cast<uint64_t*>(coefficients.data()); }
  size_t num_terms() const { return D; }
  size_t degree() const { for (size_t i = D; i-- > 0;) if (coefficients[i] != F::ZERO()) return i; return 0; }   // mod.rs:113-115
  F leading_coefficient() const { for (size_t i = D; i-- > 0;) if (coefficients[i] != F::ZERO()) return coefficients[i]; return F::ZERO(); }
  F evaluate(F x) const { F r; check(ronk_poly_eval(F::ORDER, raw(), D, x.value, &r.value)); return r; }        // mod.rs:133-139
  template <size_t D2> Polynomial<Monomial, F, D + D2> pow_mult(F coeff) const {                                  // mod.rs:153-157
    Polynomial<Monomial, F, D + D2> r; r.coefficients.fill(F::ZERO());
    std::array<F, D> c; c.fill(coeff);
    check(ronk_vec_mul(F::ORDER, raw(), reinterpret_cast<const uint64_t*>(c.data()), r.raw() + D2, D));
    return r;
  }
  Polynomial<Lagrange<F>, F, D> dft() const {                 // mod.rs:240-258: any D | ORDER-1
    std::array<F, D> out;
    check(ronk_dft(F::ORDER, F::PRIMITIVE_ELEMENT().value, raw(), reinterpret_cast<uint64_t*>(out.data()), D));
    return Polynomial<Lagrange<F>, F, D>::new_(out);
  }
  Polynomial<Lagrange<F>, F, D> fft() const {                 // mod.rs:273-292
    static_assert(D != 0 && (D & (D - 1)) == 0, "fft: D must be a power of two");  // [(); D.is_power_of_two() as usize - 1]:
    Polynomial<Lagrange<F>, F, D> r; r.basis.nodes.resize(D);
    check(ronk_fft(F::ORDER, F::PRIMITIVE_ELEMENT().value, raw(), reinterpret_cast<uint64_t*>(r.coefficients.data()),
                   reinterpret_cast<uint64_t*>(r.basis.nodes.data()), D));
    return r;
  }
  std::pair<Polynomial, Polynomial> quotient_and_remainder_dyn(const uint64_t* b, size_t d2) const {  // mod.rs:170-225
    std::pair<Polynomial, Polynomial> qr;
    check(ronk_poly_divrem(F::ORDER, raw(), D, b, d2, qr.first.raw(), qr.second.raw()));
    return qr;
  }
  friend bool operator==(const Polynomial& a, const Polynomial& b) { return a.coefficients == b.coefficients; }
  Polynomial operator-() const { Polynomial r; check(ronk_vec_neg(F::ORDER, raw(), r.raw(), D)); return r; }   // arithmetic.rs:77-95
};

template <class F, size_t D>
Polynomial<Monomial, F, D> Polynomial<Lagrange<F>, F, D>::ifft() const {
  static_assert(D != 0 && (D & (D - 1)) == 0, "ifft: D must be a power of two");
  Polynomial<Monomial, F, D> r;
  check(ronk_ifft(F::ORDER, F::PRIMITIVE_ELEMENT().value, reinterpret_cast<const uint64_t*>(coefficients.data()), r.raw(), D));
  return r;
}

// impl Add / Sub (arithmetic.rs:16-68): result has len(lhs) coefficients
template <class F, size_t D, size_t D2>
Polynomial<Monomial, F, D> operator+(const Polynomial<Monomial, F, D>& a, const Polynomial<Monomial, F, D2>& b) {
  Polynomial<Monomial, F, D> r; check(ronk_poly_add(F::ORDER, a.raw(), D, b.raw(), D2, r.raw())); return r;
}
template <class F, size_t D, size_t D2>
Polynomial<Monomial, F, D> operator-(const Polynomial<Monomial, F, D>& a, const Polynomial<Monomial, F, D2>& b) {
  Polynomial<Monomial, F, D> r; check(ronk_poly_sub(F::ORDER, a.raw(), D, b.raw(), D2, r.raw())); return r;
}
// impl Mul (arithmetic.rs:97-119): D + D2 - 1 coefficients
template <class F, size_t D, size_t D2>
Polynomial<Monomial, F, D + D2 - 1> operator*(const Polynomial<Monomial, F, D>& a, const Polynomial<Monomial, F, D2>& b) {
  Polynomial<Monomial, F, D + D2 - 1> r;
  check(ronk_poly_mul(F::ORDER, F::PRIMITIVE_ELEMENT().value, a.raw(), D, b.raw(), D2, r.raw()));
  return r;
}
// impl Div / Rem (arithmetic.rs:121-146)
template <class F, size_t D, size_t D2>
Polynomial<Monomial, F, D> operator/(const Polynomial<Monomial, F, D>& a, const Polynomial<Monomial, F, D2>& b) {
  return a.quotient_and_remainder_dyn(b.raw(), D2).first;
}
template <class F, size_t D, size_t D2>
Polynomial<Monomial, F, D> operator%(const Polynomial<Monomial, F, D>& a, const Polynomial<Monomial, F, D2>& b) {
  return a.quotient_and_remainder_dyn(b.raw(), D2).second;
}

// ---- Reed-Solomon (src/codes/reed_solomon.rs) ---------------------------------------------------------
template <class F> struct Coordinate { F x, y; };                                                   // :27-35
template <size_t N, size_t K, class F> struct Codeword { std::array<Coordinate<F>, N> data; };       // :20-25
template <size_t K, class F>
struct Message {                                                                                     // :14-18
  std::array<F, K> data;
  static Message new_(std::array<F, K> d) { return Message{d}; }                                     // :39
  // Message::encode::<N> (:42-52): x_i = root^i, y_i = polynomial.evaluate(root^i)
  template <size_t N> Codeword<N, K, F> encode() const {
    static_assert(N >= K, "Code size must be greater than or equal to K");                            // assert_ge, :108-110
    std::array<uint64_t, N> xs, ys;
    check(ronk_rs_encode(F::ORDER, F::PRIMITIVE_ELEMENT().value, reinterpret_cast<const uint64_t*>(data.data()), K, N,
                         xs.data(), ys.data()));
    Codeword<N, K, F> cw;
    for (size_t i = 0; i < N; i++) { cw.data[i].x.value = xs[i]; cw.data[i].y.value = ys[i]; }
    return cw;
  }
  // Message::decode::<M> (:54-106): interpolate the first K coordinates
  template <size_t M> static Message decode(const Codeword<M, K, F>& cw) {
    static_assert(M >= K, "Code size must be greater than or equal to K");
    std::array<uint64_t, K> xs, ys;
    for (size_t i = 0; i < K; i++) { xs[i] = cw.data[i].x.value; ys[i] = cw.data[i].y.value; }
    Message m;
    check(ronk_rs_decode(F::ORDER, xs.data(), ys.data(), K, reinterpret_cast<uint64_t*>(m.data.data())));
    return m;
  }
};

// ---- KZG commit / open (src/kzg/setup.rs:45-78) over AffinePoint<PlutoExtendedCurve> (src/curve/mod.rs:66-73) --------
struct AffinePoint {                       // Point(x, y) with x = x0 + x1 t, y = y0 + y1 t in GF(101^2), or Infinity
  std::array<uint64_t, 5> w{0, 0, 0, 0, 1};
  static AffinePoint Infinity() { return AffinePoint{}; }
  static AffinePoint new_(uint64_t x0, uint64_t x1, uint64_t y0, uint64_t y1) { AffinePoint p; p.w = {x0, x1, y0, y1, 0}; return p; }
  friend bool operator==(const AffinePoint& a, const AffinePoint& b) { return a.w == b.w; }
};
inline const ronk_curve& PlutoExtendedCurve() { static const ronk_curve c{101, 99, 0, 3}; return c; }   // pluto_curve.rs:39-51
namespace kzg {
// commit(coeffs, g1_srs): SUM g1_srs[i] * coeffs[i]
template <class S>
AffinePoint commit(const std::vector<S>& coeffs, const std::vector<AffinePoint>& g1_srs, const ronk_curve& curve = PlutoExtendedCurve()) {
  std::vector<uint64_t> pts, sc;
  for (auto& p : g1_srs) pts.insert(pts.end(), p.w.begin(), p.w.end());
  for (auto& c : coeffs) sc.push_back(c.value);
  AffinePoint out;
  check(ronk_curve_msm(&curve, pts.data(), g1_srs.size(), sc.data(), sc.size(), out.w.data()));
  return out;
}
// open::<D>(coeffs, eval_point, g1_srs): commit(poly.div([-z, 1]).coefficients, g1_srs)
template <class S, size_t D>
AffinePoint open(const std::array<S, D>& coeffs, S eval_point, const std::vector<AffinePoint>& g1_srs) {
  auto poly = Polynomial<Monomial, S, D>::new_(coeffs);
  auto divisor = Polynomial<Monomial, S, 2>::new_(std::array<S, 2>{-eval_point, S::ONE()});
  auto q = poly / divisor;
  return commit(std::vector<S>(q.coefficients.begin(), q.coefficients.end()), g1_srs);
}
}  // namespace kzg

// ---- the same commit over a production-size group: BN254 (alt_bn128) G1, y^2 = x^3 + 3 -------------------------------
// ronkathon's field traits are usize-wide (src/algebra/mod.rs:8-13), so the 254-bit types are plain data here: four 64-bit
// little-endian limbs per coordinate / scalar, standard form, (0, 0) = the point at infinity (what ronk_msm_bn254 takes).
namespace bn254 {
using Limbs = std::array<uint64_t, 4>;
struct G1Affine {
  Limbs x{0, 0, 0, 0}, y{0, 0, 0, 0};
  static G1Affine Infinity() { return G1Affine{}; }
  static G1Affine Generator() { G1Affine g; g.x = {1, 0, 0, 0}; g.y = {2, 0, 0, 0}; return g; }
  friend bool operator==(const G1Affine& a, const G1Affine& b) { return a.x == b.x && a.y == b.y; }
};
static_assert(sizeof(G1Affine) == 64, "G1Affine is passed to the C ABI as 8 words");
// kzg::commit (src/kzg/setup.rs:48-60): SUM g1_srs[i] * coeffs[i]; asserts like the reference that the SRS is long enough
inline G1Affine commit(const std::vector<Limbs>& coeffs, const std::vector<G1Affine>& g1_srs) {
  if (g1_srs.size() < coeffs.size()) throw Panic(RONK_ERR_INDEX);
  G1Affine out;
  check(ronk_msm_bn254(reinterpret_cast<const uint64_t*>(g1_srs.data()), reinterpret_cast<const uint64_t*>(coeffs.data()),
                       coeffs.size(), reinterpret_cast<uint64_t*>(&out)));
  return out;
}
// kzg::open (src/kzg/setup.rs:63-78) on the same curve: poly.div([-eval_point, ONE]) over BN254's scalar field, then commit of the
// quotient; also hands back poly(eval_point).  Same assert on the SRS length as commit.
struct Opening { G1Affine proof; Limbs value; };
inline Opening open(const std::vector<Limbs>& coeffs, const Limbs& eval_point, const std::vector<G1Affine>& g1_srs) {
  Opening o;
  check(ronk_kzg_open_bn254(reinterpret_cast<const uint64_t*>(coeffs.data()), coeffs.size(), eval_point.data(),
                            reinterpret_cast<const uint64_t*>(g1_srs.data()), g1_srs.size(), reinterpret_cast<uint64_t*>(&o.proof),
                            o.value.data()));
  return o;
}
}  // namespace bn254

// ---- heap- / device-resident polynomials, plans, the sharded transform ------------------------------------------------
// The counterpart of rust/ronk-goldilocks/src/device.rs: the reference's Polynomial<B, F, D> stores [F; D] inline
// (src/polynomial/mod.rs:34-44), which the stack cannot hold at D = 2^22; these carry the same data without a template
// length and keep it in HBM between operations.  Goldilocks (p = 2^64 - 2^32 + 1, generator 7) only.  Same values and
// panics as fft (mod.rs:273-323), ifft (:430-484), Mul (arithmetic.rs:97-119), evaluate (mod.rs:133-139), Div by a linear
// divisor (kzg::open, src/kzg/setup.rs:63-78).
namespace device {
constexpr uint64_t P = RONK_GOLDILOCKS_P, G = RONK_GOLDILOCKS_G;

class DevicePoly {
 public:
  explicit DevicePoly(size_t len) : len_(len) {
    void* p = nullptr;
    check(ronk_dev_alloc(&p, len * 8));
    ptr_ = static_cast<uint64_t*>(p);
  }
  explicit DevicePoly(const std::vector<uint64_t>& coefficients) : DevicePoly(coefficients.size()) {
    check(ronk_memcpy_h2d(ptr_, coefficients.data(), len_ * 8));
  }
  DevicePoly(DevicePoly&& o) noexcept : ptr_(o.ptr_), len_(o.len_) { o.ptr_ = nullptr; }
  DevicePoly(const DevicePoly&) = delete;
  DevicePoly& operator=(const DevicePoly&) = delete;
  ~DevicePoly() { if (ptr_) { ronk_dev_sync(); ronk_dev_free(ptr_); } }
  size_t size() const { return len_; }
  uint64_t* data() { return ptr_; }
  const uint64_t* data() const { return ptr_; }
  std::vector<uint64_t> to_host() const {
    std::vector<uint64_t> v(len_);
    check(ronk_dev_sync());
    check(ronk_memcpy_d2h(v.data(), ptr_, len_ * 8));
    return v;
  }
  DevicePoly mul(const DevicePoly& rhs) const {            // impl Mul: len + rhs.len - 1 coefficients
    DevicePoly out(len_ + rhs.len_ - 1);
    check(ronk_poly_mul_dev(P, G, ptr_, len_, rhs.ptr_, rhs.len_, out.ptr_, nullptr));
    return out;
  }
  uint64_t evaluate(uint64_t x) const {                    // Polynomial::<Monomial>::evaluate
    DevicePoly y(1);
    check(ronk_poly_eval_dev(P, ptr_, len_, x, y.ptr_, nullptr));
    return y.to_host()[0];
  }
  // self / [b0, b1] and the remainder's constant term: kzg::open's poly.div([-z, 1])
  std::pair<DevicePoly, uint64_t> div_linear(uint64_t b0, uint64_t b1) const {
    DevicePoly q(len_), r(1);
    check(ronk_poly_div_linear_dev(P, ptr_, len_, b0, b1, q.ptr_, r.ptr_, nullptr));
    const uint64_t rem = r.to_host()[0];
    return {std::move(q), rem};
  }

 private:
  uint64_t* ptr_ = nullptr;
  size_t len_ = 0;
};

// ronk_plan: twiddles + scratch for one (n = 2^log2n, batch); two_lanes = ronk_plan_opts::in_flight = 2
class Plan {
 public:
  Plan(uint32_t log2n, uint64_t batch = 1, bool two_lanes = false) : log2n_(log2n), batch_(batch) {
    ronk_plan_opts o = RONK_PLAN_OPTS_DEFAULT;
    if (two_lanes) o.in_flight = 2;
    check(ronk_plan_create_opts(&h_, P, G, log2n, batch, -1, &o));
  }
  Plan(const Plan&) = delete;
  Plan& operator=(const Plan&) = delete;
  ~Plan() { if (h_) ronk_plan_destroy(h_); }
  size_t n() const { return (size_t)1 << log2n_; }
  int in_flight() const { return ronk_plan_in_flight(h_); }
  // every polynomial handed to the plan holds exactly batch * n elements (the library sees bare pointers: a shorter
  // buffer would be read / written out of bounds) -- the Rust mirror asserts the same (rust/.../device.rs)
  void need(const DevicePoly& p) const { if (p.size() != n() * batch_) throw Panic(RONK_ERR_INVALID); }
  void forward(const DevicePoly& src, DevicePoly& dst) const { need(src); need(dst); check(ronk_ntt_forward_dev(h_, src.data(), dst.data(), nullptr)); }
  void inverse(const DevicePoly& src, DevicePoly& dst) const { need(src); need(dst); check(ronk_ntt_inverse_dev(h_, src.data(), dst.data(), nullptr)); }
  // K unrelated polynomials per call: the library keeps two transforms in flight
  void forward_many(const std::vector<const DevicePoly*>& src, const std::vector<DevicePoly*>& dst) const {
    std::vector<const uint64_t*> in;
    std::vector<uint64_t*> out;
    if (src.size() != dst.size()) throw Panic(RONK_ERR_INVALID);
    for (auto* p : src) { need(*p); in.push_back(p->data()); }
    for (auto* p : dst) { need(*p); out.push_back(p->data()); }
    check(ronk_ntt_forward_many_dev(h_, in.data(), out.data(), in.size(), nullptr));
  }
  // host vectors through the plan (a batch is pipelined over its polynomials: upload | transform | download)
  std::vector<uint64_t> forward_host(const std::vector<uint64_t>& x) const {
    if (x.size() != n() * batch_) throw Panic(RONK_ERR_INVALID);
    std::vector<uint64_t> y(x.size());
    check(ronk_ntt_forward(h_, x.data(), y.data(), nullptr));
    return y;
  }

 private:
  ronk_plan* h_ = nullptr;
  uint32_t log2n_;
  uint64_t batch_;
};

// ronk_sharded_plan: one transform over `devices` (four-step; exchange = peer-copy mesh or RCCL)
class ShardedPlan {
 public:
  ShardedPlan(uint32_t log2n, const std::vector<int>& devices, bool inverse = false, int chunks = 0, int exchange = RONK_EXCHANGE_MESH)
      : n_((size_t)1 << log2n) {
    check(ronk_sharded_plan_create_ex(&h_, log2n, inverse ? 1 : 0, devices.data(), (int)devices.size(), chunks, exchange));
  }
  // the same over any odd prime p with 2^log2n | p - 1 and primitive element g (ronk_sharded_plan_create_p)
  ShardedPlan(uint64_t p, uint64_t g, uint32_t log2n, const std::vector<int>& devices, bool inverse = false, int chunks = 0,
              int exchange = RONK_EXCHANGE_MESH)
      : n_((size_t)1 << log2n) {
    check(ronk_sharded_plan_create_p(&h_, p, g, log2n, inverse ? 1 : 0, devices.data(), (int)devices.size(), chunks, exchange));
  }
  ShardedPlan(const ShardedPlan&) = delete;
  ShardedPlan& operator=(const ShardedPlan&) = delete;
  ~ShardedPlan() { if (h_) ronk_sharded_plan_destroy(h_); }
  std::vector<uint64_t> transform(const std::vector<uint64_t>& x) const {   // natural order in and out
    if (x.size() != n_) throw Panic(RONK_ERR_INVALID);
    std::vector<uint64_t> y(n_);
    check(ronk_ntt_sharded(h_, x.data(), y.data()));
    return y;
  }

 private:
  ronk_sharded_plan* h_ = nullptr;
  size_t n_;
};
}  // namespace device

}  // namespace ronkathon
