// Reads like the reference's own tests (src/polynomial/tests.rs, src/polynomial/arithmetic.rs tests,
// src/algebra/field/prime tests) but runs every array operation on the GPU through the C ABI.
// Built and run by tests/test_cpp_host_mirror.py (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "../../ronkathon_amd/host/ronkathon.hpp"

using namespace ronkathon;
static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)
template <class Fn> static bool panics(Fn f, int code) { try { f(); } catch (const Panic& p) { return p.code == code; } return false; }
template <class F, size_t D> static std::array<F, D> arr(std::initializer_list<uint64_t> v) {
  std::array<F, D> a; size_t i = 0; for (auto x : v) a[i++] = F::new_(x); return a;
}
using B = PlutoBaseField;

int main() {
  // fixture poly(): 1 + 2x + 3x^2 + 4x^3 over F_101 (polynomial/tests.rs:3-11)
  auto poly = Polynomial<Monomial, B, 4>::new_(arr<B, 4>({1, 2, 3, 4}));
  CHECK(poly.evaluate(B::new_(2)) == B::new_(49));                                  // evaluation
  CHECK((Polynomial<Monomial, B, 3>::new_(arr<B, 3>({1, 0, 3})).evaluate(B::new_(0)) == B::new_(1)));  // evaluation_with_zero
  CHECK(panics([] { Polynomial<Monomial, B, 3>::new_(arr<B, 3>({1, 2, 3})).dft(); }, RONK_ERR_NO_ROOT));  // no_roots_of_unity
  CHECK((poly.dft().coefficients == arr<B, 4>({10, 79, 99, 18})));                  // check_coefficients / dft
  CHECK((poly.fft().coefficients == arr<B, 4>({10, 79, 99, 18})));                  // fft
  CHECK(poly.fft().ifft() == poly);                                                 // ifft round trip
  CHECK(poly.fft() == poly.dft());                                                  // same Lagrange nodes too
  CHECK(poly.dft().evaluate(B::new_(2)) == B::new_(49));                            // lagrange_evaluation
  CHECK(poly.degree() == 3);
  CHECK(poly.leading_coefficient() == B::new_(4));
  CHECK((poly.pow_mult<2>(B::new_(5)).coefficients == arr<B, 6>({0, 0, 5, 10, 15, 20})));
  // arithmetic.rs tests
  auto a = Polynomial<Monomial, B, 4>::new_(arr<B, 4>({1, 2, 3, 4}));
  auto b = Polynomial<Monomial, B, 5>::new_(arr<B, 5>({5, 6, 7, 8, 9}));
  CHECK(((b + a).coefficients == arr<B, 5>({6, 8, 10, 12, 9})));
  CHECK(((a - b).coefficients == arr<B, 4>({97, 97, 97, 97})));
  CHECK(((b - a).coefficients == arr<B, 5>({4, 4, 4, 4, 9})));
  CHECK(((-a).coefficients == arr<B, 4>({100, 99, 98, 97})));
  CHECK(((a / b).coefficients == arr<B, 4>({0, 0, 0, 0})));
  CHECK(((b / a).coefficients == arr<B, 5>({95, 78, 0, 0, 0})));
  CHECK(((a % b).coefficients == a.coefficients));
  CHECK(((b % a).coefficients == arr<B, 5>({11, 41, 71, 0, 0})));
  CHECK(((a * b).coefficients == arr<B, 8>({5, 16, 34, 60, 70, 70, 59, 36})));
  auto c = Polynomial<Monomial, B, 2>::new_(arr<B, 2>({1, 2}));
  auto d = Polynomial<Monomial, B, 2>::new_(arr<B, 2>({3, 4}));
  CHECK(((c * d).coefficients == arr<B, 3>({3, 10, 8})));
  // prime field surface
  CHECK(B::new_(40) * B::new_(61) == B::new_(16));
  CHECK(*B::new_(15).inverse() == B::new_(27));
  CHECK(!B::new_(0).inverse().has_value());
  CHECK(panics([] { (void)(B::new_(3) / B::new_(0)); }, RONK_ERR_ZERO_INVERSE));
  CHECK(B::PRIMITIVE_ELEMENT() == B::new_(2) && PlutoScalarField::PRIMITIVE_ELEMENT() == PlutoScalarField::new_(14));
  CHECK(panics([] { PlutoScalarField::primitive_root_of_unity(3); }, RONK_ERR_NO_ROOT));   // not_primitive_root_of_unity
  CHECK(panics([] { PrimeField<100>::new_(0); }, RONK_ERR_NOT_PRIME));                     // non_prime_is_not_finite_field
  // FieldExt (prime/mod.rs:142-226; rstest cases :392-408): scalar Tonelli-Shanks on the host, arrays on the GPU
  CHECK(B::new_(4).sqrt() == std::make_pair(B::new_(2), B::new_(99)));
  CHECK(B::new_(5).sqrt() == std::make_pair(B::new_(45), B::new_(56)));
  CHECK(B::new_(6).sqrt() == std::make_pair(B::new_(39), B::new_(62)));
  CHECK(B::new_(0).sqrt() == std::make_pair(B::new_(0), B::new_(0)));
  CHECK(panics([] { (void)B::new_(2).sqrt(); }, RONK_ERR_NOT_RESIDUE));
  CHECK(B::new_(4).euler_criterion() && !B::new_(2).euler_criterion());
  {
    std::vector<B> sq = {B::new_(4), B::new_(5), B::new_(6), B::new_(0)};
    auto roots = B::vec_sqrt(sq);
    CHECK((roots.first == std::vector<B>{B::new_(2), B::new_(45), B::new_(39), B::new_(0)}));
    CHECK((roots.second == std::vector<B>{B::new_(99), B::new_(56), B::new_(62), B::new_(0)}));
    CHECK((B::vec_euler({B::new_(4), B::new_(2), B::new_(0)}) == std::vector<uint64_t>{1, 0, 0}));
    CHECK(panics([] { (void)B::vec_sqrt({B::new_(4), B::new_(2)}); }, RONK_ERR_NOT_RESIDUE));
    using GF = GoldilocksField;
    GF x = GF::new_(0x123456789ABCDEFull), y = x * x;
    auto r = y.sqrt();
    CHECK(r.first * r.first == y && (r.first == x || r.second == x) && r.first.value < r.second.value);
    CHECK((GF::vec_sqrt({y}).first[0] == r.first));
  }
  // the 64-bit field: derived vector and a 2^16 round trip (too big for the stack -> heap)
  using G = GoldilocksField;
  CHECK(G::PRIMITIVE_ELEMENT() == G::new_(7));
  auto g4 = Polynomial<Monomial, G, 4>::new_(arr<G, 4>({1, 2, 3, 4}));
  CHECK((g4.fft().coefficients == arr<G, 4>({10, 18446181119461163007ull, 18446744069414584319ull, 562949953421310ull})));
  constexpr size_t N = 1 << 16;
  auto big = std::make_unique<Polynomial<Monomial, G, N>>();
  uint64_t s = 0x5EED0002;
  for (auto& v : big->coefficients) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = G::new_(s); }
  auto lag = std::make_unique<Polynomial<Lagrange<G>, G, N>>(big->fft());
  auto back = std::make_unique<Polynomial<Monomial, G, N>>(lag->ifft());
  CHECK(*back == *big);
  CHECK(lag->basis.nodes[1] == G::primitive_root_of_unity(N));
  CHECK(lag->coefficients[0] == [&] { G acc = G::ZERO(); for (auto& v : big->coefficients) acc = acc + v; return acc; }());  // X[0] = sum x_j
  // codes/reed_solomon.rs tests (:136-218) over the Mersenne prime 127
  using M7 = PrimeField<127>;
  auto msg = Message<3, M7>::new_(arr<M7, 3>({1, 2, 3}));
  auto cw3 = msg.encode<3>();
  CHECK(cw3.data[1].x == M7::new_(107) && cw3.data[2].x == M7::new_(19));
  CHECK(cw3.data[0].y == M7::new_(6) && cw3.data[1].y == M7::new_(18) && cw3.data[2].y == M7::new_(106));
  CHECK((Message<3, M7>::decode<7>(msg.encode<7>()).data == msg.data));
  auto msg5 = Message<5, M7>::new_(arr<M7, 5>({1, 2, 3, 4, 5}));
  CHECK((Message<5, M7>::decode<7>(msg5.encode<7>()).data == msg5.data));
  // kzg/tests.rs:92-176: commits over the SRS g1 * tau^i and the opening of (x-1)(x-2)(x-3) at 4
  using S = PlutoScalarField;
  std::vector<AffinePoint> srs = {AffinePoint::new_(1, 0, 2, 0), AffinePoint::new_(68, 0, 74, 0), AffinePoint::new_(65, 0, 98, 0),
                                  AffinePoint::new_(18, 0, 49, 0), AffinePoint::new_(1, 0, 99, 0), AffinePoint::new_(68, 0, 27, 0),
                                  AffinePoint::new_(65, 0, 3, 0)};
  CHECK(kzg::commit(std::vector<S>{S::new_(11), S::new_(11), S::new_(11), S::new_(1)}, srs) == AffinePoint::Infinity());
  CHECK(kzg::commit(std::vector<S>{S::new_(7), S::new_(16), S::new_(1), S::new_(11), S::new_(1)}, srs) == AffinePoint::new_(32, 0, 59, 0));
  CHECK(kzg::commit(std::vector<S>{S::new_(3), S::new_(2), S::new_(1)}, srs) == AffinePoint::new_(32, 0, 59, 0));
  CHECK((kzg::open<S, 4>(arr<S, 4>({11, 11, 11, 1}), S::new_(4), srs) == AffinePoint::new_(26, 0, 45, 0)));
  // kzg::commit over BN254 G1 (bucket-method MSM): 5*G + 7*G == 3*(4*G) == 12*G, computed three ways; G - G == infinity
  {
    using namespace bn254;
    const G1Affine g = G1Affine::Generator();
    const G1Affine g4 = commit({Limbs{4, 0, 0, 0}}, {g});
    const G1Affine a = commit({Limbs{5, 0, 0, 0}, Limbs{7, 0, 0, 0}}, {g, g});
    const G1Affine b = commit({Limbs{3, 0, 0, 0}}, {g4});
    const G1Affine c = commit({Limbs{12, 0, 0, 0}}, {g});
    CHECK(a == b && b == c && !(a == G1Affine::Infinity()));
    // r - 1 (the group order minus one) times G is -G: adding G gives infinity
    const Limbs rm1{0x43e1f593f0000000ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    CHECK(commit({rm1, Limbs{1, 0, 0, 0}}, {g, g}) == G1Affine::Infinity());
    // 2G: the EIP-196 test vector
    const G1Affine two = commit({Limbs{2, 0, 0, 0}}, {g});
    CHECK(two.x == (Limbs{0xd3c208c16d87cfd3ull, 0xd97816a916871ca8ull, 0x9b85045b68181585ull, 0x030644e72e131a02ull}));
    // kzg::open over BN254: p(x) = 1 + 2x + 3x^2 + 4x^3 at z = 2 against four copies of G: quotient [24, 11, 4, 0] -> proof 39 G, p(2) = 49
    const Opening op = open({Limbs{1, 0, 0, 0}, Limbs{2, 0, 0, 0}, Limbs{3, 0, 0, 0}, Limbs{4, 0, 0, 0}}, Limbs{2, 0, 0, 0}, {g, g, g, g});
    CHECK(op.proof == commit({Limbs{39, 0, 0, 0}}, {g}) && op.value == (Limbs{49, 0, 0, 0}));
    bool threw = false;
    try { (void)open({Limbs{1, 0, 0, 0}, Limbs{2, 0, 0, 0}}, Limbs{2, 0, 0, 0}, {g}); } catch (const Panic& e) { threw = e.code == RONK_ERR_INDEX; }
    CHECK(threw);
  }
  // device-resident polynomials, plans with two lanes, the sharded transform (ronkathon::device, the C++ twin of
  // rust/ronk-goldilocks/src/device.rs): properties only -- this test has no oracle; tests/test_gpu_parity.py and the FFI
  // replay compare the same entry points against the oracle element by element
  {
    using namespace device;
    const size_t n = (size_t)1 << 16;
    std::vector<uint64_t> x(n), y2(n);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (auto& v : x) { do { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st; } while (v >= P); }
    DevicePoly dx(x), dy(n), dz(n);
    Plan plan(16), lanes(16, 1, true);
    CHECK(plan.in_flight() == 1 && lanes.in_flight() == 2);
    plan.forward(dx, dy);
    plan.inverse(dy, dz);
    CHECK(dz.to_host() == x);                                       // ifft(fft(x)) == x
    CHECK(plan.forward_host(x) == dy.to_host());                    // host-pointer and device forms agree
    DevicePoly da(n), db(n);
    lanes.forward_many({&dx, &dz}, {&da, &db});                     // two arrays in one call, both equal to fft(x)
    CHECK(da.to_host() == dy.to_host() && db.to_host() == dy.to_host());
    // kzg::open: p = q (x - z) + r with r = p(z)
    const uint64_t z = 0x123456789ABCDEF1ull % P, t = 0xFEEDFACE12345ull;
    auto qr = dx.div_linear(P - z, 1);
    CHECK(qr.second == dx.evaluate(z));
    using F = PrimeField<RONK_GOLDILOCKS_P>;
    CHECK(F::new_(dx.evaluate(t)) == F::new_(qr.first.evaluate(t)) * (F::new_(t) - F::new_(z)) + F::new_(qr.second));
    // Mul: evaluation homomorphism
    std::vector<uint64_t> bsmall(x.begin(), x.begin() + 1000);
    DevicePoly dbs(bsmall);
    DevicePoly prod = dx.mul(dbs);
    CHECK(prod.size() == n + 999);
    CHECK(F::new_(prod.evaluate(t)) == F::new_(dx.evaluate(t)) * F::new_(dbs.evaluate(t)));
    // sharded over two logical ranks on device 0 == the single-GPU transform
    ShardedPlan sp(16, {0, 0});
    CHECK(sp.transform(x) == dy.to_host());
  }
  printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
