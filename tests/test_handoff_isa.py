"""The cross-workgroup hand-offs of the GMM kernel (the rows of a run's virtual slices -> the last
arriver; mixture state and sampler parameters -> the next waypoint's launch) rest
on properties of the EMITTED gfx950 ISA, which a compiler update could change without a test on the
GPU noticing for a long time (cdna_hip_programming.md Guideline 16).  This test disassembles the
code object inside the built libpocs.so offline (llvm-objdump; no GPU needed) and checks the shapes:

  * every store of handed-off bytes is write-through (`sc1`); the only plain global store of the
    kernel is the moments row, which leaves the launch through the kernel boundary;
  * every load of handed-off bytes bypasses L1 (`sc1`);
  * the consumer's acquire is there (`buffer_inv sc1`) and is waited for (`s_waitcnt vmcnt(0)`)
    before the barrier that releases the reading waves;
  * before a ticket / queue atomic the storing waves drain (`s_waitcnt vmcnt(0)`) and meet (`s_barrier`).
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
pytestmark = pytest.mark.skipif(not Path(OBJDUMP).exists(), reason="llvm-objdump not available")


@pytest.fixture(scope="module")
def listing(tmp_path_factory, pocs):
    lib = Path(pocs.library_path())
    assert lib.exists(), "libpocs.so not built"
    work = tmp_path_factory.mktemp("isa")
    local = work / "libpocs.so"
    shutil.copy(lib, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, cwd=work, stdout=subprocess.DEVNULL)
    co = [p for p in work.iterdir() if "gfx950" in p.name]
    assert len(co) == 1, "no gfx950 code object in libpocs.so"
    out = subprocess.run([OBJDUMP, "-d", str(co[0])], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None and ln.strip():
            cur.append(ln.split("//")[0].strip())
    return funcs


def kernel(listing, sub):
    names = [n for n in listing if sub in n]
    assert len(names) == 1, (sub, names)
    return listing[names[0]]


def mem_ops(ops, prefix):
    return [o for o in ops if o.startswith(prefix)]


@pytest.mark.parametrize("sub,streaming_bits", [("k_gmm_stepILi3ELb1ELi512ELb0E", "nt"), ("k_gmm_stepILi8ELb1ELi512ELb0E", "nt")])
def test_gmm_kernel_handoff_shapes(listing, sub, streaming_bits):
    ops = kernel(listing, sub)
    stores = mem_ops(ops, "global_store")
    # the sample stream: three 16-byte pose stores (+ one flags store) per pair of samples, in each of
    # the two forms of the iteration (whole-wave and general)
    poses = [o for o in stores if o.startswith("global_store_dwordx4")]
    assert len(poses) == 6 and all(streaming_bits in o.split() for o in poses), stores
    flags = [o for o in stores if o.startswith("global_store_dword ")]
    assert flags and all(("sc1" in o.split()) or ("nt" in o.split() and streaming_bits == "nt") for o in flags), flags
    # handed-off bytes (partial row, state, param): 8-byte write-through stores
    handed = [o for o in stores if o.startswith("global_store_dwordx2")]
    plain = [o for o in handed if "sc1" not in o.split()]
    # the plain stores: moments[w][r], which leaves through the kernel boundary -- the shard's and, after the
    # exchange between ranks (exchange_in_tail), the world's -- in each of the two written-out copies of the closer
    # (a block's first run, and the second one its range may cross into)
    assert len(handed) >= 4 and len(plain) == 4, handed
    # L1-bypassing loads of partial rows / params / state (at least one site each)
    loads_sc1 = [o for o in mem_ops(ops, "global_load_dwordx2") if "sc1" in o.split()]
    assert len(loads_sc1) >= 3, loads_sc1
    assert not mem_ops(ops, "flat_"), "flat accesses in a GMM kernel"
    # the consumer's acquire: buffer_inv sc1, waited for before the next barrier
    inv = [i for i, o in enumerate(ops) if o.startswith("buffer_inv") and "sc1" in o.split()]
    assert inv, "no agent-scope acquire in " + sub
    for i in inv:
        nxt = next(j for j in range(i + 1, len(ops)) if ops[j].startswith(("s_barrier", "s_endpgm")))
        assert any(o.startswith("s_waitcnt") and "vmcnt(0)" in o for o in ops[i + 1:nxt]), (sub, i)
    # the ticket: a returning agent-scope add behind the drain of the rows' stores and behind the barrier at
    # which every wave has drained
    atom = [i for i, o in enumerate(ops) if o.startswith("global_atomic_add")]
    assert atom, "no ticket atomic"
    seen = 0
    for i in atom:
        prev_store = max((j for j in range(i) if ops[j].startswith("global_store_dwordx2") and "sc1" in ops[j].split()), default=None)
        if prev_store is None:
            continue                                                   # the queue's dequeue ahead of any store
        between = ops[prev_store + 1:i]
        if not any(o.startswith("s_waitcnt") and "vmcnt(0)" in o for o in between):
            continue
        if not any(o.startswith("s_barrier") for o in between):
            continue
        seen += 1
    assert seen >= 1, "no sc1 store -> s_waitcnt vmcnt(0) [-> s_barrier] -> atomic sequence in " + sub


def test_advance_kernel_publishes_write_through(listing):
    ops = kernel(listing, "k_gmm_advance")
    stores = mem_ops(ops, "global_store")
    assert stores and all("sc1" in o.split() for o in stores), stores


@pytest.mark.parametrize("sub", ["k_gmm_stepILi3ELb1ELi512ELb1E", "k_gmm_stepILi8ELb0ELi512ELb1E"])
def test_lone_form_has_no_handoff_inside_the_launch(listing, sub):
    """The launch form of a lone call (one run per call) hands nothing from block to block INSIDE a launch: the rows
    and the records leave through the kernel boundary and the next launch reads them behind it.  So: no ticket
    atomic, no agent-scope acquire, no spin -- whatever a compiler update does to the stores' cache bits."""
    ops = kernel(listing, sub)
    assert not [o for o in ops if o.startswith("global_atomic")], "an atomic in the lone form"
    assert not [o for o in ops if o.startswith("buffer_inv")], "an acquire in the lone form"
    assert not mem_ops(ops, "flat_"), "flat accesses in a GMM kernel"
    assert not [o for o in ops if o.startswith("s_sleep")], "a wait loop in the lone form"


def _longest_run_of_loads(ops):
    """The longest stretch of global loads with no wait for memory (`s_waitcnt vmcnt`) inside it."""
    best = cur = 0
    for o in ops:
        if o.startswith("global_load"):
            cur += 1
            best = max(best, cur)
        elif o.startswith("s_waitcnt") and "vmcnt" in o:
            cur = 0
    return best


@pytest.mark.parametrize("sub,least", [("k_gmm_stepILi3ELb1ELi512ELb1E", 32), ("k_gmm_stepILi8ELb1ELi512ELb1E", 32),
                                       ("k_gmm_stepILi3ELb1ELi512ELb0E", 32), ("k_mc_stepILb1E", 4), ("k_mc_stepILb0E", 4)])
def test_heads_request_their_inputs_in_one_round_trip(listing, sub, least):
    """A block's head asks for everything it needs before it waits for any of it ("a head's inputs in
    ONE memory round trip"): the heads of the lone form of k_gmm_step and the closers of the batch form have a thread's 32 row
    loads in flight together (round 4's compiler had folded an item's first addition into the branch of its first load, with
    a wait in front of the other fifteen, and kept the copy loops beside them as load / wait / store: eight dependent round
    trips in a lone head); an MC block the collision world and the sector table.  Counted in the listing's linear order, so a
    lower bound of what is in flight.  A property of the emitted ISA that no GPU test would notice going."""
    ops = kernel(listing, sub)
    assert _longest_run_of_loads(ops) >= least, (sub, _longest_run_of_loads(ops))
