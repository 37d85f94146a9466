// ronk_dist.hip -- C ABI of libronk_ntt.so, part 4: the multi-GPU four-step transform (plan.h build_dist_phase1/2).
//   * ronk_dist_*      one rank's local phases; the exchange between them is the caller's all-to-all (one process per
//                      GPU: ronkathon_amd/dist.py issues it through torch.distributed = RCCL over xGMI)
//   * ronk_sharded_*   the whole sharded transform inside the library for a single-process host (the Rust host of
//                      BASELINE config 5): one compute stream and one copy stream PER PEER per device, the exchange as a
//                      mesh of hipMemcpyPeerAsync copies over xGMI -- or as RCCL send/recv groups (librccl dlopen()ed) --
//                      in column chunks so that chunk j is on the links while chunk j+1 is still being computed
//                      (SURVEY.md 8e "overlap")
#include <chrono>

#include "runtime.h"

// The field of a four-step plan: Goldilocks with the reference-convention generator 7 (the shift-twiddle kernels), or any odd
// prime p < 2^64 with n | p - 1 whose g is a quadratic non-residue -- omega_n = g^((p-1)/n) then has order exactly n and the
// phases run the same tile bodies over Montgomery arithmetic (field_policy.h MontField; plan.h HostField::montgomery), as
// ronk_plan_create does for one GPU.  The reference's transform is generic over the modulus (src/algebra/field/prime/mod.rs:
// 39-52, src/polynomial/mod.rs:273-323).  No radix-2 fallback here: a g that generates no full 2-power subgroup is
// RONK_ERR_UNSUPPORTED.
static int dist_field(u64 p, u64 g, uint32_t log2n, HostField* hf) {
  if (p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G) {
    if (log2n > 32) return RONK_ERR_NO_ROOT;   // 2-adicity of p - 1 is 32
    *hf = HostField::goldilocks();
    return RONK_OK;
  }
  if (p < 3) return RONK_ERR_INVALID;
  RCHK(ronk_check_prime(p));
  if (log2n > 63 || (p - 1) % ((u64)1 << log2n) != 0) return RONK_ERR_NO_ROOT;   // field/mod.rs:72
  if (h_powmod(g % p, (p - 1) / 2, p) != p - 1) return RONK_ERR_UNSUPPORTED;
  *hf = HostField::montgomery(p, g);
  return RONK_OK;
}

// ------------------------------------------------------------------------------ multi-GPU four-step
struct ronk_dist_plan {
  DistShape sh;
  CompiledPlan p1, p2;
  u64* d_tmp = nullptr;  // n / world elements
  int device;
  int chunks = 1;        // column chunks of phase 1 (plan.h); p1 is compiled for chunk 0
};
extern "C" int ronk_dist_plan_destroy(ronk_dist_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->p1.release(); pl->p2.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  delete pl;
  return RONK_OK;
}
extern "C" int ronk_dist_plan_create_p(ronk_dist_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse, int rank,
                                       int world, int device, int chunks);
extern "C" int ronk_dist_plan_create(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device) {
  return ronk_dist_plan_create_p(out, RONK_GOLDILOCKS_P, RONK_GOLDILOCKS_G, log2n, inverse, rank, world, device, 1);
}
extern "C" int ronk_dist_plan_create_chunked(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world,
                                             int device, int chunks) {
  return ronk_dist_plan_create_p(out, RONK_GOLDILOCKS_P, RONK_GOLDILOCKS_G, log2n, inverse, rank, world, device, chunks);
}
extern "C" int ronk_dist_plan_create_p(ronk_dist_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse, int rank,
                                       int world, int device, int chunks) {
  if (!out || world < 1 || rank < 0 || rank >= world) return RONK_ERR_INVALID;
  *out = nullptr;
  HostField hf;
  RCHK(dist_field(p, g, log2n, &hf));
  DistShape sh;
  if (!dist_shape((int)log2n, world, &sh)) return RONK_ERR_UNSUPPORTED;
  if (!dist_chunks_ok(sh, chunks)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_dist_plan* pl = new ronk_dist_plan();
  pl->sh = sh; pl->device = device; pl->chunks = chunks;
  int rc = pl->p1.compile(build_dist_phase1((int)log2n, inverse != 0, rank, world, 4, 0, 0, chunks, hf));
  if (!rc) rc = pl->p2.compile(build_dist_phase2((int)log2n, inverse != 0, rank, world, 4, 0, chunks, hf));
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, (sh.n / sh.W) * 8);
    if (e == hipSuccess) e = hipStreamSynchronize(0);   // table uploads (null-stream copies) before any non-blocking stream uses them
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  if (rc) { ronk_dist_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}
// one column chunk of phase 1: reads columns [chunk*Cwc, (chunk+1)*Cwc) of d_in ([R][C/world]) and writes the contiguous
// piece d_send[chunk*R*Cwc ..) = `world` blocks [R/world][Cwc]
extern "C" int ronk_dist_phase1_chunk_dev(ronk_dist_plan* pl, int chunk, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send || chunk < 0 || chunk >= pl->chunks) return RONK_ERR_INVALID;
  const u64 Cwc = pl->sh.Cw / (u64)pl->chunks;
  return pl->p1.run(d_in + (u64)chunk * Cwc, nullptr, d_send + (u64)chunk * pl->sh.R * Cwc, pl->d_tmp, (hipStream_t)st,
                    ~(u64)0, ~(u64)0, 0, (u64)chunk * Cwc);
}
extern "C" int ronk_dist_phase1_dev(ronk_dist_plan* pl, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send) return RONK_ERR_INVALID;
  for (int j = 0; j < pl->chunks; j++) RCHK(ronk_dist_phase1_chunk_dev(pl, j, d_in, d_send, st));
  return RONK_OK;
}
extern "C" int ronk_dist_phase2_dev(ronk_dist_plan* pl, const uint64_t* d_recv, uint64_t* d_out, void* st) {
  if (!pl || !d_recv || !d_out || d_recv == d_out) return RONK_ERR_INVALID;
  return pl->p2.run(d_recv, nullptr, d_out, pl->d_tmp, (hipStream_t)st);
}


// ------------------------------------------------------------------------------ sharded transform, single process
// Rank g = devices[g].  Per rank: phase-1 plan (compiled for chunk 0; chunk j shifts the twiddle's column offset),
// phase-2 plan, scratch, send and receive buffers (n/W elements each), a compute stream and ONE COPY STREAM PER PEER:
// the W copies of a chunk leave on W streams, so all xGMI links of the device (point-to-point, one per peer) carry data
// at the same time instead of one copy queue feeding them one after the other.
//   compute_g   :  phase 1 chunk 0 | chunk 1 | ... | (wait: every sender's copies to g have landed) phase 2
//   copy_g[h]   :        (wait chunk 0) block for h | (wait chunk 1) block for h | ...
// Buffers are reused by the next call: the copies of call k+1 into rank h wait for phase 2 of call k on h (recv), and
// phase 1 of call k+1 waits for the copies of call k out of its own device (send) -- all by events, nothing blocks the host.
//
// Exchange = RONK_EXCHANGE_RCCL: the same chunk schedule with the W x W blocks of a chunk as ONE RCCL group of
// ncclSend / ncclRecv pairs (an all-to-all over xGMI the way north_star words it), on one communication stream per device.
// librccl.so is dlopen()ed on first use -- the library has no link-time dependency on it -- so a host can A/B the peer-copy
// mesh against RCCL per plan.  Ranks must then sit on distinct devices (one RCCL rank per GPU).
#include <dlfcn.h>

namespace {
typedef void* rcclComm_t;
struct RcclApi {
  void* handle = nullptr;
  int (*CommInitAll)(rcclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int kNcclUint64 = 5;   // ncclDataType_t::ncclUint64 (rccl.h)
std::mutex g_rccl_mu;
RcclApi g_rccl;
std::string g_rccl_err;

// loads librccl once; false (and a message for ronk_last_hip_error) if it is not there or lacks a symbol
bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  if (g_rccl.handle) return false;   // tried before, incomplete
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) { g_rccl_err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found"); return false; }
  auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
  g_rccl.CommInitAll = (int (*)(rcclComm_t*, int, const int*))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (int (*)(rcclComm_t))sym("ncclCommDestroy");
  g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
  g_rccl.Send = (int (*)(const void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclSend");
  g_rccl.Recv = (int (*)(void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclRecv");
  g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.Send && g_rccl.Recv;
  if (!g_rccl.ok) g_rccl_err = "librccl.so lacks one of ncclCommInitAll / ncclGroupStart / ncclSend / ncclRecv";
  return g_rccl.ok;
}
}  // namespace

int rccl_fail(int code, const char* what);   // ronk_core.hip: records the text, returns RONK_ERR_RCCL
#define RCCLCHK(call)                                    \
  do {                                                   \
    int r_ = (call);                                     \
    if (r_ != 0) return rccl_fail(r_, #call);            \
  } while (0)

struct ShardRank {
  int device = 0;
  CompiledPlan p1, p2;
  u64 *tmp = nullptr, *send = nullptr, *recv = nullptr;
  u64 *stage_in = nullptr, *stage_out = nullptr;   // host-pointer entry point only (lazy)
  hipStream_t compute = nullptr;
  std::vector<hipStream_t> copy;                   // [W]: the stream that carries blocks for rank h (mesh; one per destination
                                                   // DEVICE) / [1]: the RCCL stream
  std::vector<bool> owns;                          // copy[h] was created for h (not an alias of an earlier entry)
  std::vector<hipEvent_t> chunk_done;              // [chunks]
  std::vector<hipEvent_t> sent_to;                 // [W]: this rank's blocks for rank h have left (mesh) / [1] (RCCL)
  hipEvent_t done = nullptr;                       // phase 2 has read the receive buffer
  rcclComm_t comm = nullptr;
  bool used = false;
};
struct ronk_sharded_plan {
  DistShape sh;
  int ndev = 0, chunks = 1;
  bool inverse = false;
  int exchange = RONK_EXCHANGE_MESH;
  std::vector<ShardRank> r;
  // how a block travels from rank g to rank h, [g * ndev + h] (ronk_sharded_plan_peer_access): RONK_PEER_SAME_DEVICE,
  // RONK_PEER_DIRECT (peer access enabled: xGMI / PCIe peer-to-peer) or RONK_PEER_STAGED (the runtime refused peer access:
  // hipMemcpyPeerAsync then goes through host memory -- correct, an order of magnitude slower, and never silent)
  std::vector<int> peer;
  std::mutex mu;
};

static int on_device(int dev) {
  HIPCHK(hipSetDevice(dev));
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_destroy(ronk_sharded_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  for (auto& k : pl->r) {
    (void)hipSetDevice(k.device);
    if (k.compute) (void)hipStreamSynchronize(k.compute);
    for (size_t h = 0; h < k.copy.size(); h++) if (k.copy[h] && k.owns[h]) (void)hipStreamSynchronize(k.copy[h]);
  }
  for (auto& k : pl->r) {
    (void)hipSetDevice(k.device);
    if (k.comm && g_rccl.ok) (void)g_rccl.CommDestroy(k.comm);
    k.p1.release(); k.p2.release();
    for (u64* q : {k.tmp, k.send, k.recv, k.stage_in, k.stage_out}) if (q) (void)hipFree(q);
    for (auto e : k.chunk_done) if (e) (void)hipEventDestroy(e);
    for (auto e : k.sent_to) if (e) (void)hipEventDestroy(e);
    if (k.done) (void)hipEventDestroy(k.done);
    if (k.compute) (void)hipStreamDestroy(k.compute);
    for (size_t h = 0; h < k.copy.size(); h++) if (k.copy[h] && k.owns[h]) (void)hipStreamDestroy(k.copy[h]);
  }
  delete pl;
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_create(ronk_sharded_plan** out, uint32_t log2n, int inverse, const int* devices, int ndev,
                                        int chunks) {
  return ronk_sharded_plan_create_ex(out, log2n, inverse, devices, ndev, chunks, RONK_EXCHANGE_MESH);
}

extern "C" int ronk_sharded_plan_create_p(ronk_sharded_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse,
                                          const int* devices, int ndev, int chunks, int exchange);
extern "C" int ronk_sharded_plan_create_ex(ronk_sharded_plan** out, uint32_t log2n, int inverse, const int* devices, int ndev,
                                           int chunks, int exchange) {
  return ronk_sharded_plan_create_p(out, RONK_GOLDILOCKS_P, RONK_GOLDILOCKS_G, log2n, inverse, devices, ndev, chunks, exchange);
}
extern "C" int ronk_sharded_plan_create_p(ronk_sharded_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse,
                                          const int* devices, int ndev, int chunks, int exchange) {
  if (!out || !devices || ndev < 1) return RONK_ERR_INVALID;
  if (exchange != RONK_EXCHANGE_MESH && exchange != RONK_EXCHANGE_RCCL) return RONK_ERR_INVALID;
  *out = nullptr;
  HostField hf;
  RCHK(dist_field(p, g, log2n, &hf));
  DistShape sh;
  if (!dist_shape((int)log2n, ndev, &sh)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  int ndevices = 0;
  HIPCHK(hipGetDeviceCount(&ndevices));
  for (int g = 0; g < ndev; g++)
    if (devices[g] < 0 || devices[g] >= ndevices) return RONK_ERR_INVALID;
  if (exchange == RONK_EXCHANGE_RCCL) {
    for (int g = 0; g < ndev; g++)
      for (int h = 0; h < g; h++)
        if (devices[g] == devices[h]) return RONK_ERR_UNSUPPORTED;   // one RCCL rank per GPU
    if (!rccl_load()) return rccl_fail(-1, g_rccl_err.c_str());
  }
  if (chunks <= 0) {   // default: up to 4 chunks (the exchange of chunk j hides under chunks j+1 ..)
    chunks = 4;
    while (chunks > 1 && !dist_chunks_ok(sh, chunks)) chunks >>= 1;
  }
  if (!dist_chunks_ok(sh, chunks)) return RONK_ERR_UNSUPPORTED;
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  ronk_sharded_plan* pl = new ronk_sharded_plan();
  pl->sh = sh; pl->ndev = ndev; pl->chunks = chunks; pl->inverse = inverse != 0; pl->exchange = exchange;
  pl->r.resize(ndev);
  pl->peer.assign((size_t)ndev * ndev, RONK_PEER_SAME_DEVICE);
  const size_t per = (size_t)(sh.n / sh.W);
  const int ncopy = exchange == RONK_EXCHANGE_RCCL ? 1 : ndev;
  int rc = RONK_OK;
  for (int g = 0; g < ndev && !rc; g++) {
    ShardRank& k = pl->r[g];
    k.device = devices[g];
    rc = on_device(k.device);
    // peer access to every other device of the plan (xGMI); "already enabled" is fine.  A refusal (hipDeviceCanAccessPeer
    // says no, or enabling fails) leaves hipMemcpyPeerAsync staging through host memory: the outcome is KEPT per pair and
    // reported (ronk_sharded_plan_peer_access; bench.py prints it), and RONK_REQUIRE_PEER=1 turns it into an error at plan
    // creation.  RONK_FORCE_NO_PEER=1 (tests) takes the refused branch without asking the runtime; RONK_FORCE_NO_PEER=ranks
    // (tests on a ONE-GPU box) additionally treats two different logical ranks that share a device as a refused pair, so that
    // the report, RONK_REQUIRE_PEER and the hipMemcpyPeerAsync route (sharded_enqueue) are exercised there as well.
    static const char* const fnp = getenv("RONK_FORCE_NO_PEER");
    static const bool force_no_peer = fnp != nullptr, force_ranks = fnp && !strcmp(fnp, "ranks");
    for (int h = 0; h < ndev && !rc; h++) {
      int& how = pl->peer[(size_t)g * ndev + h];
      if (devices[h] == k.device && !(force_ranks && h != g)) { how = RONK_PEER_SAME_DEVICE; continue; }
      int can = 0;
      hipError_t pe = force_no_peer ? hipErrorPeerAccessUnsupported : hipDeviceCanAccessPeer(&can, k.device, devices[h]);
      if (pe == hipSuccess && can) {
        pe = hipDeviceEnablePeerAccess(devices[h], 0);
        if (pe == hipErrorPeerAccessAlreadyEnabled) pe = hipSuccess;
      } else if (pe == hipSuccess) {
        pe = hipErrorPeerAccessUnsupported;
      }
      (void)hipGetLastError();
      how = pe == hipSuccess ? RONK_PEER_DIRECT : RONK_PEER_STAGED;
    }
    if (!rc) rc = k.p1.compile(build_dist_phase1((int)log2n, inverse != 0, g, ndev, 4, 0, 0, chunks, hf));
    if (!rc) rc = k.p2.compile(build_dist_phase2((int)log2n, inverse != 0, g, ndev, 4, 0, chunks, hf));
    if (!rc && (k.p1.pd.passes.empty() || k.p2.pd.passes.empty())) rc = RONK_ERR_UNSUPPORTED;
    hipError_t e = hipSuccess;
    if (!rc) e = hipStreamSynchronize(0);   // the table uploads (null-stream copies) before the plan's non-blocking streams use them
    if (!rc) e = hipMalloc((void**)&k.tmp, per * 8);
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&k.send, per * 8);
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&k.recv, per * 8);
    if (!rc && e == hipSuccess) e = hipStreamCreateWithFlags(&k.compute, hipStreamNonBlocking);
    // mesh: one copy stream per destination DEVICE (= per xGMI link, plus one for blocks that stay on this device); logical
    // ranks that share a GPU share the stream -- there is one link to keep busy, and every extra stream costs queue switches
    k.copy.assign(ncopy, nullptr); k.sent_to.assign(ncopy, nullptr); k.owns.assign(ncopy, false);
    for (int h = 0; h < ncopy && !rc && e == hipSuccess; h++) {
      int same = -1;
      if (exchange != RONK_EXCHANGE_RCCL)
        for (int h2 = 0; h2 < h && same < 0; h2++) if (devices[h2] == devices[h]) same = h2;   // an earlier rank on that device
      if (same >= 0) k.copy[h] = k.copy[same];
      else { e = hipStreamCreateWithFlags(&k.copy[h], hipStreamNonBlocking); k.owns[h] = e == hipSuccess; }
      if (e == hipSuccess) e = hipEventCreateWithFlags(&k.sent_to[h], hipEventDisableTiming);
    }
    k.chunk_done.resize(chunks, nullptr);
    for (int j = 0; j < chunks && !rc && e == hipSuccess; j++) e = hipEventCreateWithFlags(&k.chunk_done[j], hipEventDisableTiming);
    if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&k.done, hipEventDisableTiming);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "sharded plan resources");
  }
  if (!rc && getenv("RONK_REQUIRE_PEER")) {
    for (int v : pl->peer)
      if (v == RONK_PEER_STAGED) { rc = RONK_ERR_UNSUPPORTED; g_rccl_err = "peer access refused between two devices of the plan (RONK_REQUIRE_PEER)"; break; }
  }
  if (!rc && exchange == RONK_EXCHANGE_RCCL) {
    std::vector<rcclComm_t> comms(ndev, nullptr);
    const int r_ = g_rccl.CommInitAll(comms.data(), ndev, devices);
    if (r_ != 0) rc = rccl_fail(r_, "ncclCommInitAll");
    else for (int g = 0; g < ndev; g++) pl->r[g].comm = comms[g];
  }
  (void)hipSetDevice(prev);
  if (rc) { ronk_sharded_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_exchange(const ronk_sharded_plan* pl) { return pl ? pl->exchange : RONK_ERR_INVALID; }

extern "C" int ronk_sharded_plan_peer_access(const ronk_sharded_plan* pl, int* matrix, int capacity) {
  if (!pl) return RONK_ERR_INVALID;
  const int w = pl->ndev;
  if (matrix && capacity < w * w) return RONK_ERR_INVALID;
  int staged = 0;
  for (int i = 0; i < w * w; i++) {
    if (matrix) matrix[i] = pl->peer[i];
    if (pl->peer[i] == RONK_PEER_STAGED) staged++;
  }
  return staged;
}

extern "C" int ronk_sharded_plan_info(const ronk_sharded_plan* pl, uint64_t* rows, uint64_t* cols, uint64_t* per_rank,
                                      int* chunks) {
  if (!pl) return RONK_ERR_INVALID;
  if (rows) *rows = pl->sh.R;
  if (cols) *cols = pl->sh.C;
  if (per_rank) *per_rank = pl->sh.n / pl->sh.W;
  if (chunks) *chunks = pl->chunks;
  return RONK_OK;
}

static int sharded_enqueue(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out) {
  const DistShape& sh = pl->sh;
  const int W = pl->ndev, chunks = pl->chunks;
  const bool rccl = pl->exchange == RONK_EXCHANGE_RCCL;
  const u64 Cwc = sh.Cw / (u64)chunks, blk = sh.Rw * Cwc;
  // buffer reuse across calls
  for (int g = 0; g < W; g++) {
    ShardRank& k = pl->r[g];
    if (!k.used) continue;
    RCHK(on_device(k.device));
    for (auto ev : k.sent_to) HIPCHK(hipStreamWaitEvent(k.compute, ev, 0));     // send buffer: the previous call's blocks are out
    if (rccl) { for (int h = 0; h < W; h++) HIPCHK(hipStreamWaitEvent(k.copy[0], pl->r[h].done, 0)); }   // a group writes every recv buffer
    else for (int h = 0; h < W; h++) HIPCHK(hipStreamWaitEvent(k.copy[h], pl->r[h].done, 0));           // recv of h: its phase 2 has read it
  }
  // phase 1 + exchange.  Chunk-major (every device gets its chunk 0 before any device gets chunk 1: the host's enqueue
  // time is spread over the devices, and an RCCL group sees every rank's chunk j at once); rank-major only when all the
  // ranks are logical ranks on ONE device, where interleaving them just multiplies the cross-stream waits.
  bool one_device = !rccl;
  for (int g = 1; g < W; g++) one_device = one_device && pl->r[g].device == pl->r[0].device;
  const int outer = one_device ? W : chunks, inner = one_device ? chunks : W;
  for (int o = 0; o < outer; o++) {
    for (int i = 0; i < inner; i++) {
      const int j = one_device ? i : o, g = one_device ? o : i;
      ShardRank& k = pl->r[g];
      RCHK(on_device(k.device));
      u64* piece = k.send + (u64)j * sh.R * Cwc;
      RCHK(k.p1.run(d_in[g] + (u64)j * Cwc, nullptr, piece, k.tmp, k.compute, ~(u64)0, ~(u64)0, 0, (u64)j * Cwc));
      HIPCHK(hipEventRecord(k.chunk_done[j], k.compute));
      if (rccl) { HIPCHK(hipStreamWaitEvent(k.copy[0], k.chunk_done[j], 0)); continue; }
      for (int hh = 0; hh < W; hh++) {
        const int h = (g + hh) % W;                                     // start with the local block, then ring order
        u64* dst = pl->r[h].recv + ((u64)g * chunks + j) * blk;
        const u64* src = piece + (u64)h * blk;
        HIPCHK(hipStreamWaitEvent(k.copy[h], k.chunk_done[j], 0));
        if (pl->peer[(size_t)g * W + h] == RONK_PEER_SAME_DEVICE) HIPCHK(hipMemcpyAsync(dst, src, blk * 8, hipMemcpyDeviceToDevice, k.copy[h]));
        else HIPCHK(hipMemcpyPeerAsync(dst, pl->r[h].device, src, k.device, blk * 8, k.copy[h]));
      }
    }
    if (rccl) {
      const int j = o;
      // one group: rank g sends block h of its chunk to rank h and receives block (h, j) from rank h
      RCCLCHK(g_rccl.GroupStart());
      int r_ = 0;
      for (int g = 0; g < W && !r_; g++) {
        ShardRank& k = pl->r[g];
        const u64* piece = k.send + (u64)j * sh.R * Cwc;
        for (int h = 0; h < W && !r_; h++) {
          r_ = g_rccl.Send(piece + (u64)h * blk, (size_t)blk, kNcclUint64, h, k.comm, k.copy[0]);
          if (!r_) r_ = g_rccl.Recv(k.recv + ((u64)h * chunks + j) * blk, (size_t)blk, kNcclUint64, h, k.comm, k.copy[0]);
        }
      }
      const int r2 = g_rccl.GroupEnd();
      if (r_) return rccl_fail(r_, "ncclSend / ncclRecv");
      if (r2) return rccl_fail(r2, "ncclGroupEnd");
    }
  }
  for (int g = 0; g < W; g++) {
    ShardRank& k = pl->r[g];
    RCHK(on_device(k.device));
    for (size_t h = 0; h < k.copy.size(); h++) HIPCHK(hipEventRecord(k.sent_to[h], k.copy[h]));
  }
  // phase 2: rank h waits for what was sent TO IT (mesh: one event per sender; RCCL: every rank's communication stream)
  for (int h = 0; h < W; h++) {
    ShardRank& k = pl->r[h];
    RCHK(on_device(k.device));
    for (int g = 0; g < W; g++) HIPCHK(hipStreamWaitEvent(k.compute, pl->r[g].sent_to[rccl ? 0 : h], 0));
    RCHK(k.p2.run(k.recv, nullptr, d_out[h], k.tmp, k.compute));
    HIPCHK(hipEventRecord(k.done, k.compute));
    k.used = true;
  }
  return RONK_OK;
}

// Diagnostics for the first runs on a real node (bench.py --workload sharded): ONE transform in three SERIALISED stages -- every
// rank's phase 1 (all chunks), the whole exchange, every rank's phase 2 -- with all devices drained between the stages, wall
// time per stage in ms[0..2].  The product path (sharded_enqueue) overlaps the stages chunk by chunk; this one exists so that a
// slow result can be attributed: stage times, and with them the achieved rate per directed link
// (n * 8 / W^2 bytes per ordered rank pair / ms[1]).  Same results in d_out as ronk_ntt_sharded_dev.
static int sharded_drain(ronk_sharded_plan* pl) {
  for (auto& k : pl->r) {
    RCHK(on_device(k.device));
    for (size_t h = 0; h < k.copy.size(); h++) if (k.owns[h]) HIPCHK(hipStreamSynchronize(k.copy[h]));
    HIPCHK(hipStreamSynchronize(k.compute));
  }
  return RONK_OK;
}
static int sharded_timed(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out, float* ms) {
  const DistShape& sh = pl->sh;
  const int W = pl->ndev, chunks = pl->chunks;
  const bool rccl = pl->exchange == RONK_EXCHANGE_RCCL;
  const u64 Cwc = sh.Cw / (u64)chunks, blk = sh.Rw * Cwc;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  RCHK(sharded_drain(pl));
  auto t0 = now();
  for (int j = 0; j < chunks; j++)
    for (int g = 0; g < W; g++) {
      ShardRank& k = pl->r[g];
      RCHK(on_device(k.device));
      RCHK(k.p1.run(d_in[g] + (u64)j * Cwc, nullptr, k.send + (u64)j * sh.R * Cwc, k.tmp, k.compute, ~(u64)0, ~(u64)0, 0, (u64)j * Cwc));
    }
  RCHK(sharded_drain(pl));
  ms[0] = since(t0);
  t0 = now();
  for (int j = 0; j < chunks; j++) {
    if (rccl) {
      RCCLCHK(g_rccl.GroupStart());
      int r_ = 0;
      for (int g = 0; g < W && !r_; g++) {
        ShardRank& k = pl->r[g];
        const u64* piece = k.send + (u64)j * sh.R * Cwc;
        for (int h = 0; h < W && !r_; h++) {
          r_ = g_rccl.Send(piece + (u64)h * blk, (size_t)blk, kNcclUint64, h, k.comm, k.copy[0]);
          if (!r_) r_ = g_rccl.Recv(k.recv + ((u64)h * chunks + j) * blk, (size_t)blk, kNcclUint64, h, k.comm, k.copy[0]);
        }
      }
      const int r2 = g_rccl.GroupEnd();
      if (r_) return rccl_fail(r_, "ncclSend / ncclRecv");
      if (r2) return rccl_fail(r2, "ncclGroupEnd");
      continue;
    }
    for (int g = 0; g < W; g++) {
      ShardRank& k = pl->r[g];
      RCHK(on_device(k.device));
      const u64* piece = k.send + (u64)j * sh.R * Cwc;
      for (int hh = 0; hh < W; hh++) {
        const int h = (g + hh) % W;
        u64* dst = pl->r[h].recv + ((u64)g * chunks + j) * blk;
        const u64* src = piece + (u64)h * blk;
        if (pl->peer[(size_t)g * W + h] == RONK_PEER_SAME_DEVICE) HIPCHK(hipMemcpyAsync(dst, src, blk * 8, hipMemcpyDeviceToDevice, k.copy[h]));
        else HIPCHK(hipMemcpyPeerAsync(dst, pl->r[h].device, src, k.device, blk * 8, k.copy[h]));
      }
    }
  }
  RCHK(sharded_drain(pl));
  ms[1] = since(t0);
  t0 = now();
  for (int h = 0; h < W; h++) {
    ShardRank& k = pl->r[h];
    RCHK(on_device(k.device));
    RCHK(k.p2.run(k.recv, nullptr, d_out[h], k.tmp, k.compute));
  }
  RCHK(sharded_drain(pl));
  ms[2] = since(t0);
  return RONK_OK;
}
extern "C" int ronk_sharded_time_stages(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out, float* ms) {
  if (!pl || !d_in || !d_out || !ms) return RONK_ERR_INVALID;
  for (int g = 0; g < pl->ndev; g++)
    if (!d_in[g] || !d_out[g] || d_in[g] == d_out[g]) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  int rc = sharded_timed(pl, d_in, d_out, ms);
  (void)hipSetDevice(prev);
  return rc;
}

extern "C" int ronk_ntt_sharded_dev(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out) {
  if (!pl || !d_in || !d_out) return RONK_ERR_INVALID;
  for (int g = 0; g < pl->ndev; g++)
    if (!d_in[g] || !d_out[g] || d_in[g] == d_out[g]) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  int rc = sharded_enqueue(pl, d_in, d_out);
  (void)hipSetDevice(prev);
  return rc;
}

extern "C" int ronk_sharded_sync(ronk_sharded_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  int rc = RONK_OK;
  for (auto& k : pl->r) {
    bool ok = hipSetDevice(k.device) == hipSuccess;
    for (size_t h = 0; h < k.copy.size(); h++) if (k.owns[h]) ok = ok && hipStreamSynchronize(k.copy[h]) == hipSuccess;
    ok = ok && hipStreamSynchronize(k.compute) == hipSuccess;
    if (!ok) { rc = hip_fail(hipGetLastError(), "ronk_sharded_sync"); break; }
  }
  (void)hipSetDevice(prev);
  return rc;
}

// Host natural-order vector in, natural-order vector out (Polynomial::fft / ifft of src/polynomial/mod.rs:273-323,
// :430-484 at a size sharded over the node): scatter the column blocks (strided H2D copies), transform, gather
// X[k1 + R*k2] from the ranks' [C][R/W] blocks.
extern "C" int ronk_ntt_sharded(ronk_sharded_plan* pl, const uint64_t* in, uint64_t* out) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  const DistShape& sh = pl->sh;
  const size_t per = (size_t)(sh.n / sh.W);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  std::vector<const uint64_t*> din(pl->ndev);
  std::vector<uint64_t*> dout(pl->ndev);
  int rc = RONK_OK;
  {
    std::lock_guard<std::mutex> lk(pl->mu);
    for (int g = 0; g < pl->ndev && !rc; g++) {
      ShardRank& k = pl->r[g];
      rc = on_device(k.device);
      hipError_t e = hipSuccess;
      if (!rc && !k.stage_in) e = hipMalloc((void**)&k.stage_in, per * 8);
      if (!rc && e == hipSuccess && !k.stage_out) e = hipMalloc((void**)&k.stage_out, per * 8);
      // rank g's [R][C/W] column block of the R x C input
      if (!rc && e == hipSuccess)
        e = hipMemcpy2DAsync(k.stage_in, sh.Cw * 8, in + (size_t)g * sh.Cw, sh.C * 8, sh.Cw * 8, sh.R, hipMemcpyHostToDevice, k.compute);
      if (!rc && e != hipSuccess) rc = hip_fail(e, "scatter");
      din[g] = k.stage_in; dout[g] = k.stage_out;
    }
    if (!rc) rc = sharded_enqueue(pl, din.data(), dout.data());
    for (int h = 0; h < pl->ndev && !rc; h++) {
      ShardRank& k = pl->r[h];
      rc = on_device(k.device);
      // rank h's [C][R/W] block holds X[(h*R/W + k1l) + R*k2]
      hipError_t e = hipSuccess;
      if (!rc) e = hipMemcpy2DAsync(out + (size_t)h * sh.Rw, sh.R * 8, k.stage_out, sh.Rw * 8, sh.Rw * 8, sh.C, hipMemcpyDeviceToHost, k.compute);
      if (!rc && e == hipSuccess) e = hipStreamSynchronize(k.compute);
      if (!rc && e != hipSuccess) rc = hip_fail(e, "gather");
    }
  }
  (void)hipSetDevice(prev);
  return rc;
}
