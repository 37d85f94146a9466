"""Runs the HIP tile kernel's source (ronkathon_amd/csrc/ntt_tile.h + plan.h) under the host
fiber emulator (tests/emu/emu_tile.cpp) against the oracle.  Checks the plan algebra -- strides,
digit maps, twiddle exponents, LDS swizzle, the multi-GPU phase split -- on the CPU.  The
emulator is test infrastructure: the product library does not contain it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "emu_tile")


@pytest.fixture(scope="module")
def emu():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "emu", "emu_tile.cpp")]
    deps = srcs + [os.path.join(ROOT, "ronkathon_amd", "csrc", f) for f in ("ntt_tile.h", "ntt_small.h", "ntt_mul.h", "plan.h", "gl64.h", "tile_cfg_table.h",
                                                                                  "field_policy.h", "mont64.h", "ntt_tile_wl.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        obj = os.path.join(ROOT, "build", "orc_emu.o")
        _build(["gcc", "-O2", "-c", "-o", obj, os.path.join(ROOT, "oracle", "ronk_oracle.c")], obj)
        _build(["g++", "-O2", "-std=c++17", "-o", EXE, srcs[0], obj], EXE)
    return EXE



def _build(cmd_prefix, out):
    """compile to a private name, then rename: several pytest-xdist workers may decide to rebuild the same emulator at once, and a
    binary that is being written cannot be executed ("text file busy")"""
    tmp = "%s.tmp.%d" % (out, os.getpid())
    subprocess.check_call(cmd_prefix[:cmd_prefix.index("-o") + 1] + [tmp] + cmd_prefix[cmd_prefix.index("-o") + 2:])
    os.replace(tmp, out)


def run(emu, *args, env=None):
    out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env) if env else None)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("OK"), out.stdout[-400:] + out.stderr[-400:]
    return out.stdout


# Generic odd primes on the SAME kernel source (field_policy.h MontField: Montgomery products, table-form twiddles): a prime
# above 2^63 (sums carry into bit 64), one below 2^62, a 32-bit one.  (p, primitive element) -- the GPU suite's list.
MONT_PRIMES = [(0xFFFFFFFC00000001, 10), (29 * 2**57 + 1, 3), (3 * 2**30 + 1, 5)]


def mont_env(p, g):
    return {"RONK_EMU_P": hex(p), "RONK_EMU_G": str(g)}


@pytest.mark.parametrize("p,g", MONT_PRIMES)
def test_montgomery_primes_on_the_tile_kernels(emu, p, g):
    """single-pass (staged I/O, ragged tiles), two-pass (latency form, tile form with the specialised column / row shapes and
    the full twiddle matrix, batched, inverse with the folded n^-1), implicit padding + second operand + truncation (the
    polynomial multiply's features on the generic body), three passes -- every output against the oracle called with (p, g)"""
    e = mont_env(p, g)
    for k in (4, 5, 8, 12):
        run(emu, k, 3, 0, 4, env=e)
        run(emu, k, 37, 1, 0, env=e)
    run(emu, 13, 2, 0, 4, env=e)
    out = run(emu, 16, 2, 1, 4, 18, env=e)
    assert "cfg:column/matrix" in out and "cfg:row" in out      # the specialised shapes run for Montgomery primes too
    out = run(emu, 16, 1, 0, 4, 18, 25, 0, 0, 1, env=e)
    assert "kernel=small" in out                                 # the planner's latency form
    run(emu, 15, 2, 0, 4, 18, 25, 20000, 30000, 0, 9000, 1, env=e)   # in_valid / out_valid / in_valid1 / in2
    run(emu, 14, 1, 1, 2, 0, 13, env=e)                              # three passes (4, 5, 5)
    if p > 2**63:   # the multiply's specialised feature shapes (tile_kernels_mont_feat.hip): padded pair forward, in2 + truncation inverse
        out = run(emu, 20, 2, 0, 2, 18, 25, 300000, 0, 0, 7, env=e)
        assert "feat:column" in out
        out = run(emu, 20, 1, 1, 2, 18, 25, 0, 700001, 0, 0, 1, env=e)
        assert "feat:column" in out and "feat:row" in out


def test_montgomery_fused_multiply_middle(emu):
    """ntt_mul.h over a Montgomery prime (tile_kernels_mont_mul.hip): row pass of both operands, product of canonical values as
    two Montgomery products, inverse column pass; both inverse twiddle forms, ragged lengths"""
    p, g = MONT_PRIMES[0]
    run(emu, "mul", 22, 2097157, 2097148, 2, 18, env=mont_env(p, g))
    run(emu, "mul", 21, 700001, 900000, 2, 21, env=mont_env(p, g))


def test_montgomery_dist_phases(emu):
    """the four-step phase builders take the field as well (plan.h build_dist_phase1 / 2 with a HostField)"""
    p, g = MONT_PRIMES[0]
    run(emu, "dist", 16, 4, 0, 0, 2, env=mont_env(p, g))
    run(emu, "dist", 16, 2, 1, 18, 1, env=mont_env(p, g))


@pytest.mark.parametrize("k", [4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_single_pass(emu, k):
    run(emu, k, 3, 0, 4)
    run(emu, k, 2, 1, 4)
    if k <= 6:      # tiny transforms: staged I/O (n <= 32) with full, several and ragged tiles, both tile-width rules
        run(emu, k, 200, 0, 4)
        run(emu, k, 257, 1, 0)


@pytest.mark.parametrize("k,batch,inv,logc", [(13, 2, 0, 4), (16, 1, 1, 4), (17, 1, 0, 3), (20, 1, 0, 4), (22, 1, 0, 4)])
def test_two_pass(emu, k, batch, inv, logc):
    run(emu, k, batch, inv, logc)


@pytest.mark.parametrize("k", [20, 21, 22, 23])
@pytest.mark.parametrize("logc", [0, 1, 2, 3])
def test_tuned_tile_widths(emu, k, logc):
    """ronk_plan_create_tuned(tile_log2_columns = c): bench.py times c = 2 plans on two streams; 2^23 is a three-pass plan"""
    run(emu, k, 1, (k + logc) & 1, logc, 0, 23)


WL_ENVS = [{}, {"RONK_WL_HALF": "1"}, {"RONK_WL": "0"}, {"RONK_WL": "2"}, {"RONK_WL": "3"}, {"RONK_TWF_T": "1"}]
# every knob at the sizes whose passes have 2^11 rows (where the half image / the transposed matrix exist); the other row counts with
# the default and with one pass at a time (the suite runs serially in the driver: the 2^24 case costs 15 s per run)
WL_CASES = ([(e, k, i, t) for e in WL_ENVS for (k, i, t) in ((22, 0, 22), (22, 1, 18))] +
            [(e, k, i, t) for e in ({}, {"RONK_WL": "2"}, {"RONK_WL": "3"}) for (k, i, t) in ((21, 1, 21), (23, 0, 0), (20, 0, 20), (20, 1, 18))] +
            [({"RONK_WL_HALF": "1"}, 21, 1, 21), ({"RONK_WL_HALF": "1"}, 23, 0, 0), ({}, 24, 1, 0)])


@pytest.mark.parametrize("env,k,inv,twf", WL_CASES)
def test_wave_local_tile_bodies(emu, env, k, inv, twf):
    """ntt_tile_wl.h -- the 2^10 / 2^11 / 2^12-row x 4-column column / row passes with one wave-local and one cross-wave exchange
    (round 6; the library's default for these shapes): FULL image (one barrier per pass) and, for 2^11 rows, the half image (two
    32-bit phases, two barriers), forward / inverse, two-level tables / full matrix (also transposed, RONK_TWF_T), one pass at a
    time mixed with ntt_tile.h's kernels (RONK_WL = 2 / 3), and RONK_WL=0 -- the same plans on the old kernels; the plans are
    2^10 x 2^10, 2^11 x 2^10, 2^11 x 2^11, 2^12 x 2^11 and 2^12 x 2^12.  The fibers run one after the other, so a missing
    barrier / wave_sync between dependent LDS accesses shows up as a mismatch."""
    out = run(emu, k, 1, inv, 2, twf, 25, env=env)
    kernels = [l.split("kernel=")[1] for l in out.strip().splitlines() if l.startswith("pass")]
    wl = env.get("RONK_WL", "1")
    want_col, want_row = wl in ("1", "2"), wl in ("1", "3")
    assert kernels[0].startswith("wl:column") == want_col and kernels[1].startswith("wl:row") == want_row, out
    # the same bodies over a Montgomery prime (FULL image)
    if not env and k in (20, 22):
        p, g = MONT_PRIMES[0]
        out = run(emu, k, 1, inv, 2, twf, 25, env=mont_env(p, g))
        assert "kernel=wl:column" in out and "kernel=wl:row" in out


@pytest.mark.parametrize("k,d,d2,logc,twf", [(20, 1 << 19, 1 << 19, 2, 18), (20, 300001, 7, 2, 20), (20, 1, 1 << 20, 2, 18),
                                             (22, 1 << 21, 1 << 21, 2, 18), (22, (1 << 21) + 5, (1 << 21) - 4, 2, 22),
                                             (21, 1 << 20, 1 << 20, 2, 18), (21, 700001, 900000, 2, 21),
                                             (23, (1 << 22) + 1, (1 << 22) - 104, 2, 18)])
def test_fused_multiply_middle(emu, k, d, d2, logc, twf):
    """ntt_mul.h: the multiply's forward row pass of both operands + product + inverse column pass as ONE tile body (what
    ronk_plan.hip conv_dev launches at 2^21 / 2^22 / 2^23), every product coefficient against the oracle; ragged operand lengths,
    both inverse twiddle forms (two-level tables / full matrix)"""
    run(emu, "mul", k, d, d2, logc, twf)


def test_specialised_kernels_are_selected_and_generic_bodies_still_match(emu):
    """the hot two-pass shapes run the compile-time-specialised bodies (tile_cfg_table.h; same selection rule as
    tile_kernels.hip); with RONK_NO_CFG_KERNELS the generic body computes the same plans"""
    for args, kinds in (((22, 1, 0, 4), ("cfg:column/two-level", "cfg:row")), ((22, 1, 1, 2), ("wl:column/two-level", "wl:row")),
                        ((16, 3, 0, 4, 18), ("cfg:column/matrix", "cfg:row")), ((18, 1, 1, 4, 18), ("cfg:column/matrix", "cfg:row"))):
        out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
        lines = out.stdout.strip().splitlines()
        assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
        assert [l.split("kernel=")[1] for l in lines if l.startswith("pass")] == list(kinds), out.stdout
        env = dict(os.environ, RONK_NO_CFG_KERNELS="1")
        out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
        lines = out.stdout.strip().splitlines()
        assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
        assert all(l.endswith("kernel=generic") for l in lines if l.startswith("pass"))


@pytest.mark.parametrize("k,batch", [(19, 32), (20, 16), (22, 4)])
def test_planner_many_tiles_branch(emu, k, batch):
    """build_plan(auto_tiles): >= 4 large tiles per CU -> 8192-coefficient tiles for the 2^10 / 2^11-row passes (the
    library default for big batches)"""
    run(emu, k, batch, 0, 4, 0, 23, 0, 0, 1)


def test_three_pass_batched(emu):
    run(emu, 23, 3, 0, 4, 0, 23)
    run(emu, 23, 2, 1, 3, 0, 23)


@pytest.mark.parametrize("k,batch,inv", [(13, 2, 0), (16, 2, 1), (18, 1, 0)])
def test_two_pass_full_twiddle_matrix(emu, k, batch, inv):
    """plans whose inter-pass twiddle is the full matrix laid out like the output tile (plan.h maybe_full_table)"""
    run(emu, k, batch, inv, 4, 24)


@pytest.mark.parametrize("args", [(10, 1, 0, 4, 0, 25, 300, 0), (10, 1, 1, 4, 0, 25, 0, 777), (16, 1, 0, 4, 18, 25, 30000, 0),
                                  (16, 1, 1, 4, 18, 25, 0, 65535), (13, 1, 0, 4, 18, 13, 5000, 8000)])
def test_implicit_padding_and_truncation(emu, args):
    """TileArgs::in_valid / out_valid (polynomial multiply: operands read in place as zero-padded, product truncated)"""
    run(emu, *args)


def test_dist_phases(emu):
    for k, w, inv in ((12, 1, 0), (13, 2, 0), (16, 4, 1), (20, 8, 0)):
        run(emu, "dist", k, w, inv)
    run(emu, "dist", 16, 4, 0, 24)
    # column-chunked exchange layout (ronk_sharded_*, ronk_dist_plan_create_chunked): emu_tile dist k W inv twf chunks
    for k, w, inv, chunks in ((16, 4, 0, 2), (16, 2, 1, 4), (20, 8, 0, 4), (20, 2, 1, 8)):
        run(emu, "dist", k, w, inv, 0, chunks)


def test_gl64_field_arithmetic_edges():
    """ronkathon_amd/csrc/gl64.h on the host against 128-bit arithmetic: every carry/borrow branch, all 96 shifts"""
    src = os.path.join(ROOT, "tests", "emu", "test_gl64_host.cpp")
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    compilers = ["g++"]
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        compilers.append("/opt/rocm/lib/llvm/bin/clang++")   # exercises the __builtin_subc path of the device build
    for i, cxx in enumerate(compilers):
        exe = os.path.join(ROOT, "build", "test_gl64_host_%d" % i)
        subprocess.check_call([cxx, "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ALL OK" in out.stdout, (cxx, out.stdout[-800:])


@pytest.mark.parametrize("args", [(22, 1, 0, 3), (22, 1, 1, 2), (16, 3, 0, 4, 18), (18, 1, 1, 4, 18), (20, 2, 0, 3), (19, 1, 1, 4)])
def test_half_lds_exchange(emu, args):
    """TileCfg::HALF (RONK_HALF_LDS=1: exchanges between rounds as two 32-bit phases through a half-size LDS image): the
    specialised two- and three-round bodies compute the same plans; a missing barrier between the phases shows up here
    because the fibers run one after the other"""
    env = dict(os.environ, RONK_HALF_LDS="1", RONK_WL="0")   # (the 2^11-row x 4-column passes: ntt_tile.h's kernels, not ntt_tile_wl.h's)
    out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
    assert all("kernel=half:" in l for l in lines if l.startswith("pass")), out.stdout


@pytest.mark.parametrize("args", [(13, 2, 0), (13, 1, 1), (14, 3, 1), (15, 1, 0), (16, 1, 0), (16, 4, 1), (17, 2, 0), (18, 1, 1), (17, 4, 0),
                                  (13, 128, 1), (14, 64, 0)])
def test_small_latency_kernel(emu, args):
    """ntt_small.h (4 coefficients per work-item, radix-4 rounds in place; a radix-2 round for odd pass sizes): the plans
    the planner builds (auto_tiles = 1) for at most 2^18 coefficients in all, for 2^19 when n <= 2^17 and for 2^20 when
    n <= 2^14; forward and inverse, batched"""
    k, batch, inv = args
    out = subprocess.run([emu, str(k), str(batch), str(inv), "4", "18", "25", "0", "0", "1"], capture_output=True, text=True, timeout=600)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
    assert all("kernel=small" in l for l in lines if l.startswith("pass")), out.stdout


@pytest.mark.parametrize("args", [(19, 1, 0), (18, 2, 1), (16, 16, 0), (17, 8, 0), (15, 32, 1)])
def test_planner_leaves_the_latency_form_where_the_tile_kernels_are_faster(emu, args):
    """plan.h's rule, re-measured late in round 3 (profiles/r03_small_kernel_loads.txt): one 2^19 transform, 2 x 2^18 and
    2^20 coefficients of transforms longer than 2^14 run the tile kernels"""
    k, batch, inv = args
    out = subprocess.run([emu, str(k), str(batch), str(inv), "4", "18", "25", "0", "0", "1"], capture_output=True, text=True, timeout=900)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
    assert not any("kernel=small" in l for l in lines if l.startswith("pass")), out.stdout


def test_small_kernel_fused_multiply_arguments(emu):
    """implicit zero padding on load (in_valid) and truncation on store (out_valid) through the small kernel"""
    for args in ((13, 1, 0, 4, 18, 25, 3000, 0, 1), (13, 1, 1, 4, 18, 25, 0, 5000, 1), (16, 2, 0, 4, 18, 25, 40000, 0, 1)):
        run(emu, *args)


def test_small_kernel_every_load_up_front(emu):
    """the latency kernel issues all of a pass's global loads before it waits for any (ntt_small.h): the paths whose loads used to
    sit behind branches -- two-level inter-pass twiddles (matrix limit below the size), the fused multiply's second operand,
    a second operand together with padding limits that differ per batch entry, odd pass sizes -- against the oracle"""
    for args in ((16, 1, 0, 4, 12, 25, 0, 0, 1), (15, 2, 1, 4, 10, 25, 0, 0, 1), (14, 2, 0, 4, 18, 25, 9000, 0, 1, 5000, 1),
                 (16, 1, 1, 4, 18, 25, 0, 0, 1, 0, 1), (13, 2, 0, 4, 10, 25, 3000, 0, 1, 0, 1)):
        out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
        lines = out.stdout.strip().splitlines()
        assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
        assert all("kernel=small" in l for l in lines if l.startswith("pass")), out.stdout


@pytest.mark.parametrize("k,batch,want_tiles", [(20, 1, 256), (21, 1, 256), (21, 2, 128), (22, 1, 256)])
def test_planner_narrows_tiles_until_every_cu_has_one(emu, k, batch, want_tiles):
    """auto_tiles: a pass of fewer than 256 workgroups gets narrower tiles (not below 4 columns); the specialised kernels
    still cover the resulting shapes, and the transform is still the oracle's"""
    out = subprocess.run([emu, str(k), str(batch), "0", "4", "18", "25", "0", "0", "1"], capture_output=True, text=True, timeout=900)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
    passes = [l for l in lines if l.startswith("pass")]
    assert len(passes) == 2
    for l in passes:
        f = dict(t.split("=") for t in l.split()[1:])
        assert int(f["tiles"]) == want_tiles and int(f["grid"]) == want_tiles * batch, l
        assert f["kernel"].startswith("cfg:") or f["kernel"].startswith("wl:"), l


@pytest.mark.parametrize("args", [(14, 2, 0, 4, 18, 25, 3000, 0, 1, 5000), (20, 2, 0, 2, 18, 25, 300000, 0, 0, 700001),
                                  (13, 2, 0, 4, 18, 25, 8000, 0, 0, 17)])
def test_paired_multiply_operands(emu, args):
    """TileArgs::in_valid1: a batch of two whose entries have different zero-padding limits (the two operands of a
    polynomial multiply transformed by one pair of launches) -- small kernel, tile kernel with 4-column tiles, tile kernel"""
    run(emu, *args)


@pytest.mark.parametrize("k,dirs", [(23, (0, 1)), (24, (1,))])
def test_three_pass_plans_run_the_specialised_bodies(emu, k, dirs):
    """2^23 .. : column pass (two-level twiddle) / middle pass (full matrix, one transform per row of the first split) /
    last pass (flat rows) all match a TileCfg shape now that nb2 is a run-time value there (2^25: the same shapes with 2^9
    rows -- run on the GPU, too slow for the emulator in a CPU suite)"""
    for inv in dirs:
        out = subprocess.run([emu, str(k), "1", str(inv), "4", "18", "23"], capture_output=True, text=True, timeout=1800)
        lines = out.stdout.strip().splitlines()
        assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
        kinds = [l.split("kernel=")[1] for l in lines if l.startswith("pass")]
        assert kinds == ["cfg:column/two-level", "cfg:column/matrix", "cfg:row"], out.stdout


@pytest.mark.parametrize("args,kinds", [
    ((20, 2, 0, 2, 18, 25, 300000, 0, 0, 700001), ("feat:column/two-level", "wl:row")),           # multiply: both operands, zero padded
    ((22, 2, 0, 2, 18, 25, 2097152, 0, 0, 2097152), ("feat:column/two-level", "wl:row")),
    ((20, 1, 1, 4, 18, 25, 0, 1000001, 1, 0, 1), ("feat:column/two-level", "feat:row")),            # its inverse: fused product in, truncated out
    ((21, 1, 1, 4, 18, 25, 0, 2097151, 1, 0, 1), ("feat:column/two-level", "feat:row")),
    ((22, 1, 1, 4, 18, 25, 0, 4194303, 1, 0, 1), ("feat:column/two-level", "feat:row")),
    ((16, 16, 0, 4, 18, 25, 32768, 0, 1), ("feat:column/matrix", "cfg:row")),                       # batched Reed-Solomon encode
])
def test_feature_kernels(emu, args, kinds):
    """TileCfg::FEAT: the passes of a polynomial multiply and of a batched encode on compile-time-specialised bodies (byte-scaled
    padding / truncation limits, second operand through the narrow addressing)"""
    out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=1800)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-600:]
    assert [l.split("kernel=")[1] for l in lines if l.startswith("pass")] == list(kinds), out.stdout


@pytest.mark.parametrize("args,kinds", [
    ((22, 1, 0, 4, 22, 25, 0, 0, 1), ("cfg:column/matrix", "cfg:row")),                        # the library default at 2^22 / 2^21
    ((21, 1, 1, 4, 21, 25, 0, 0, 1), ("wl:column/matrix", "cfg:row")),                         # (narrowed to 4-column tiles: ntt_tile_wl.h)
    ((22, 2, 0, 2, 22, 25, 2097152, 0, 0, 2097152), ("feat:column/matrix", "wl:row")),         # ... and the multiply on it
    ((22, 1, 1, 4, 22, 25, 0, 4194303, 1, 0, 1), ("feat:column/matrix", "feat:row")),
    ((21, 1, 1, 4, 21, 25, 0, 2097151, 1, 0, 1), ("feat:column/matrix", "feat:row")),
])
def test_full_twiddle_matrix_at_2_21_and_2_22(emu, args, kinds):
    """ronk_plan_create's default from round 3: the whole inter-pass twiddle matrix at 2^21 / 2^22 (KIND 3 shapes), also under
    the multiply's features"""
    out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=1800)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-600:]
    assert [l.split("kernel=")[1] for l in lines if l.startswith("pass")] == list(kinds), out.stdout


@pytest.mark.parametrize("args", [(12, 3, 0, 0), (12, 2, 1, 0), (11, 5, 1, 0), (10, 7, 0, 0), (9, 3, 1, 0), (8, 13, 0, 0), (7, 9, 1, 0), (6, 33, 0, 0)])
def test_single_pass_plans_run_the_whole_polynomial_body(emu, args):
    """n = 2^6 .. 2^12 at the planner's own tile widths (max_logc = 0, as ronk_plan_create passes it): KIND 5, forward and
    inverse (run-time scale), ragged last tiles"""
    out = subprocess.run([emu] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1].startswith("OK"), out.stdout[-400:]
    assert [l.split("kernel=")[1] for l in lines if l.startswith("pass")] == ["cfg:whole"], out.stdout


# ---- the linear-division kernels (ronkathon_amd/csrc/lindiv_kernels.h) on fibers: tests/emu/emu_scan.cpp
SCAN_EXE = os.path.join(ROOT, "build", "emu_scan")


@pytest.fixture(scope="module")
def emu_scan():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    src = os.path.join(ROOT, "tests", "emu", "emu_scan.cpp")
    deps = [src] + [os.path.join(ROOT, "ronkathon_amd", "csrc", f) for f in ("lindiv_kernels.h", "gl64.h")]
    if not os.path.exists(SCAN_EXE) or any(os.path.getmtime(d) > os.path.getmtime(SCAN_EXE) for d in deps):
        obj = os.path.join(ROOT, "build", "orc_emu.o")
        _build(["gcc", "-O2", "-c", "-o", obj, os.path.join(ROOT, "oracle", "ronk_oracle.c")], obj)
        _build(["g++", "-O2", "-std=c++17", "-o", SCAN_EXE, src, obj], SCAN_EXE)
    return SCAN_EXE


@pytest.mark.parametrize("direct", [0, 1])
def test_linear_division_kernels_on_fibers(emu_scan, direct):
    """both launches of ronk_poly_div_linear_dev's default path, the code the GPU runs, against the oracle's recurrence (and
    orc_poly_divrem up to 4096 coefficients): one coefficient .. several hundred chunks (more than one sum per lane in the
    carry), chunk and wavefront edges, z = 0 / 1 / p - 1, non-monic divisors, F_101 and F_2; with 16-byte reads of the
    lanes' runs and through the LDS image"""
    GP = 0xFFFFFFFF00000001
    for d in (1, 2, 7, 511, 512, 513, 2047, 2048, 2049, 4096, 4097, 20000, 70001):
        run(emu_scan, GP, d, 123456789, 1, direct)
    run(emu_scan, GP, 9000, GP - 1, 77, direct)
    run(emu_scan, GP, 9000, 1, 1, direct)
    run(emu_scan, GP, 9000, 0, 5, direct)
    run(emu_scan, 101, 5000, 7, 3, direct)
    run(emu_scan, 101, 300, 0, 1, direct)
    run(emu_scan, 2, 3000, 1, 1, direct)
    run(emu_scan, GP, 700001, 987654321987, 3, direct, 5)       # 342 chunks: two sums per lane for the low chunks


@pytest.mark.parametrize("pl", [4, 8])
@pytest.mark.parametrize("mode", [2, 3, 6])
def test_one_launch_linear_division_on_fibers(emu_scan, mode, pl):
    """lindiv_one_body -- ronk_poly_div_linear_dev's default up to 2^23 coefficients from round 6: ONE launch, the chunk kept in
    registers while the chunk sums travel through the look-back array (workgroups of 1024 lanes x 4 or 8 coefficients, run in dispatch
    order here; suffix sums of values weighted by powers of z: additions only); mode 2 / 3 = through the LDS image / 16-byte direct
    loads, 6 = every look-back wait fails, so every workgroup recomputes the chunk sums above it (the path a timeout takes on the
    device).  Lane, wavefront and chunk edges, several hundred chunks (more than 256 and more than 512: the Y^t table's third
    factor), z = 1 / p - 1 (z = 0 stays with the two launches, as in the library), non-monic divisors, F_101 and F_2; the array of
    the next call must come back cleared."""
    GP = 0xFFFFFFFF00000001
    env = {"EMU_LINDIV1_PL": str(pl)}
    for d in (1, 7, 8, 9, 513, 4096 * pl // 4 - 1, 4096 * pl // 4, 4096 * pl // 4 + 1, 70001):
        run(emu_scan, GP, d, 123456789, 1, mode, env=env)
    run(emu_scan, GP, 9000, GP - 1, 77, mode, env=env)
    run(emu_scan, GP, 9000, 1, 1, mode, env=env)
    run(emu_scan, GP, 9000, 0, 5, mode, env=env)
    run(emu_scan, 101, 25000, 7, 3, mode, env=env)
    run(emu_scan, 101, 300, 0, 1, mode, env=env)
    run(emu_scan, 2, 9000, 1, 1, mode, env=env)
    if mode == 3:
        run(emu_scan, GP, 4096 * pl // 4 * 601 + 5, 987654321987, 3, mode, 5, env=env)   # 601 chunks


# ---- the long-division kernel (ronkathon_amd/csrc/longdiv_kernel.h) on fibers: tests/emu/emu_longdiv.cpp
LONGDIV_EXE = os.path.join(ROOT, "build", "emu_longdiv")


@pytest.fixture(scope="module")
def emu_longdiv():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    src = os.path.join(ROOT, "tests", "emu", "emu_longdiv.cpp")
    deps = [src] + [os.path.join(ROOT, "ronkathon_amd", "csrc", f) for f in ("longdiv_kernel.h", "gl64.h")] + \
           [os.path.join(ROOT, "oracle", "ronk_oracle.c")]
    if not os.path.exists(LONGDIV_EXE) or any(os.path.getmtime(d) > os.path.getmtime(LONGDIV_EXE) for d in deps):
        obj = os.path.join(ROOT, "build", "orc_emu_longdiv.o")
        _build(["gcc", "-O2", "-c", "-o", obj, os.path.join(ROOT, "oracle", "ronk_oracle.c")], obj)
        _build(["g++", "-O2", "-std=c++17", "-o", LONGDIV_EXE, src, obj], LONGDIV_EXE)
    return LONGDIV_EXE


def test_long_division_kernel_on_fibers(emu_longdiv):
    """quotient_and_remainder (src/polynomial/mod.rs:170-225) as the one-workgroup kernel runs it -- the dividend copied and the
    status cleared inside the launch -- against orc_poly_divrem, value for value and panic for panic: random shapes with ragged
    divisors (trailing zeros: the reference's early stop or index panic), the zero divisor, zero and all-(p-1) dividends,
    divisors longer than the dividend, in place and out of place; Goldilocks, F_101, F_17, F_2 and a 64-bit prime on the
    generic operations; 1, 64, 256 and 1024 work-items"""
    for p, cases, maxd, items, seed in ((0xFFFFFFFF00000001, 400, 200, 64, 3), (101, 400, 200, 64, 3), (17, 400, 200, 64, 3),
                                        (2, 400, 200, 64, 3), (0xFFFFFFFFFFFFFFC5, 400, 200, 64, 3),
                                        (0xFFFFFFFF00000001, 12, 1500, 256, 9), (101, 200, 50, 1, 4), (17, 40, 90, 1024, 5)):
        out = subprocess.run([emu_longdiv, hex(p), str(cases), str(maxd), str(items), str(seed)], capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0 and out.stdout.startswith("OK"), (p, out.stdout[-300:])


@pytest.mark.parametrize("args", [(20, 1, 1, 3), (19, 2, 1, 4, 18), (18, 1, 0, 4, 18), (21, 1, 0, 2, 21)])
def test_r4_round_structure(emu, args):
    """TileCfg::R4 (tile_kernels_r4.hip, opt-in RONK_R4MID=1): passes of 2^9 / 2^10 rows as [16 . 4] . [8 | 16] -- a wave-uniform
    shift layer omega_64^(a k1) after the first round, ONE table twiddle after the 4-point round -- column passes with both
    inter-pass twiddle forms, row passes, forward and inverse, 4 / 8 / 16-column tiles; every output against the oracle"""
    out = run(emu, *args, env={"RONK_R4MID": "1"})
    assert "kernel=r4:" in out
