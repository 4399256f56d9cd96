"""The cache-maintenance instructions the peer exchange's cross-device correctness rests on, asserted in the BUILT code object
(DESIGN.md section 6 "memory model").  No GPU needed: the gfx950 code object is taken out of
libfreud_sae.so and disassembled with the toolchain's llvm-objdump.

What must hold for every barrier of p2p_allreduce_kernel<4> (fp32 / bf16 gradient segments) and <2> (fp64 statistics), and for
the statistics push inside finalize_losses_kernel:
  (barriers 0 and 1; barrier 2 only says "I have finished reading your buffer" and needs neither)
  release side:  buffer_wbl2 sc0 sc1  (dirty L2 lines -> memory)  ... s_waitcnt vmcnt(0) ...  then the first flag store, and that
                 store is a system-scope one (sc0 sc1);
  acquire side:  the flag poll is a system-scope load (sc0 sc1) in a loop with s_sleep;  buffer_inv sc0 sc1  (L1 + the non-local
                 lines of L2 dropped)  ... s_waitcnt vmcnt(0) ...  before the s_barrier that releases the workgroup's loads.
A compiler or source change that drops one of them fails here instead of on an 8-GPU node."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "freud_amd", "lib", "libfreud_sae.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

KERNELS = {
    "p2p4": "_Z20p2p_allreduce_kernelILi4EEv7P2PArgs",
    "p2p2": "_Z20p2p_allreduce_kernelILi2EEv7P2PArgs",
    "finalize": "_Z22finalize_losses_kernelPKfiS0_iPfS1_lifS0_iPKd9StatsPush",
    "push_selftest": "_Z24p2p_selftest_push_kernel9StatsPushiPj",
    # fused forward, the shipped instantiations (T = bf16 / fp16 / fp32 activations; PAD = false / true; STAMP = false)
    "ff2_bf16": "_Z22fwd_fused2_d384_kernelIDF16bLb0ELb0EEv12FwdFusedArgs",
    "ff2_bf16_pad": "_Z22fwd_fused2_d384_kernelIDF16bLb1ELb0EEv12FwdFusedArgs",
    "ff2_f16": "_Z22fwd_fused2_d384_kernelIDF16_Lb0ELb0EEv12FwdFusedArgs",
    "ff2_f32": "_Z22fwd_fused2_d384_kernelIfLb0ELb0EEv12FwdFusedArgs",
}


@pytest.fixture(scope="module")
def disasm():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: build first (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    tmp = tempfile.mkdtemp(prefix="freud_codeobj_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert len(cos) == 1, f"expected one gfx950 code object, found {cos}"
        out = {}
        for key, sym in KERNELS.items():
            r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f"--disassemble-symbols={sym}", os.path.join(tmp, cos[0])],
                               check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            ins = []
            for line in r.stdout.splitlines():
                line = line.split("//")[0].strip()
                if not line or line.endswith(":") or line.startswith(("/", "Disassembly")) or "file format" in line:
                    continue
                ins.append(re.sub(r"\s+", " ", line))
            assert len(ins) > 50, f"{sym}: not found in the code object (renamed? update KERNELS)"
            out[key] = ins
        yield out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _is_store(i):
    return i.startswith(("global_store", "flat_store", "buffer_store", "global_atomic", "flat_atomic"))


def _is_wait0(i):
    return i.startswith("s_waitcnt") and "vmcnt(0)" in i


def _check_release(ins, min_count):
    idx = [k for k, i in enumerate(ins) if i.startswith("buffer_wbl2")]
    assert len(idx) >= min_count, f"{len(idx)} buffer_wbl2, expected >= {min_count}"
    for k in idx:
        assert "sc0" in ins[k] and "sc1" in ins[k], f"release fence is not system scope: {ins[k]}"
        waited = False
        for j in range(k + 1, len(ins)):
            if _is_wait0(ins[j]):
                waited = True
            if _is_store(ins[j]):
                assert waited, f"a store follows buffer_wbl2 without s_waitcnt vmcnt(0): {ins[k:j + 1]}"
                assert "sc0" in ins[j] and "sc1" in ins[j], f"the flag store after the release is not system scope: {ins[j]}"
                break
        else:
            raise AssertionError("no store after a release fence")


def _check_acquire(ins, min_count, need_barrier=True):
    idx = [k for k, i in enumerate(ins) if i.startswith("buffer_inv")]
    assert len(idx) >= min_count, f"{len(idx)} buffer_inv, expected >= {min_count}"
    for k in idx:
        assert "sc0" in ins[k] and "sc1" in ins[k], f"acquire fence is not system scope: {ins[k]}"
        if not need_barrier:
            continue
        waited = False
        for j in range(k + 1, len(ins)):
            if _is_wait0(ins[j]):
                waited = True
            if ins[j].startswith("s_barrier"):
                assert waited, f"s_barrier follows buffer_inv without s_waitcnt vmcnt(0): {ins[k:j + 1]}"
                break
            assert not (ins[j].startswith(("global_load", "buffer_load")) and "sc1" not in ins[j] and "lds" not in ins[j] and not waited), \
                f"a plain load is issued before the invalidate has completed: {ins[k:j + 1]}"
        else:
            raise AssertionError("no s_barrier after an acquire fence")


def _check_poll(ins, min_count):
    polls = [k for k, i in enumerate(ins) if i.startswith(("global_load_dwordx2", "flat_load_dwordx2")) and "sc0" in i and "sc1" in i]
    assert len(polls) >= min_count, f"{len(polls)} system-scope 8-byte polls, expected >= {min_count}"
    assert sum(1 for i in ins if i.startswith("s_sleep")) >= min_count
    assert sum(1 for i in ins if i.startswith("s_memrealtime")) >= 2 * min_count      # every poll loop is bounded by the timeout


@pytest.mark.parametrize("key", ["p2p4", "p2p2"])
def test_exchange_kernel_barriers_release_and_acquire_at_system_scope(disasm, key):
    ins = disasm[key]
    _check_release(ins, 2)          # barriers 0 and 1 (barrier 2 publishes nothing and nothing is read behind it: no fences)
    _check_acquire(ins, 2)
    _check_poll(ins, 3)
    # every storing wave drains its stores before the workgroup barrier that precedes a release
    for k, i in enumerate(ins):
        if i.startswith("buffer_wbl2"):
            back = ins[max(0, k - 40):k]
            assert any(b.startswith("s_barrier") for b in back), "no workgroup barrier before the release fence"


@pytest.mark.parametrize("key", ["finalize", "push_selftest"])
def test_statistics_push_orders_payload_before_epoch(disasm, key):
    """The push moves three 8-byte words through UNCACHED memory with system-scope accesses: no cache maintenance, but order --
    payload stores, s_waitcnt vmcnt(0), epoch store; epoch poll (system-scope load in a bounded loop), then the payload loads."""
    ins = disasm[key]
    st = [k for k, i in enumerate(ins) if i.startswith(("global_store_dwordx2", "flat_store_dwordx2")) and "sc0" in i and "sc1" in i]
    assert len(st) >= 3, f"expected the triple as three system-scope 8-byte stores, found {len(st)}"
    # between the second payload store and the epoch store (the third system-scope store) there is a full drain
    assert any(_is_wait0(i) for i in ins[st[1] + 1:st[2]]), ins[st[0]:st[2] + 1]
    _check_poll(ins, 1)
    ld = [k for k, i in enumerate(ins) if i.startswith(("global_load_dwordx2", "flat_load_dwordx2")) and "sc0" in i and "sc1" in i]
    assert len(ld) >= 3, "epoch poll + two payload loads, all system scope"


# ---------------------------------------------------------------------------------------------------------------------
# Register spills: NO kernel of the shipped code object may touch scratch memory.  Round 4 shipped two
# instantiations of topk_select_reg_kernel with spills (836 and 116 scratch instructions); round 5 removed them -- template-constant
# loops instead of `#pragma unroll` (which gives up silently above LLVM's size threshold and leaves register arrays on the stack),
# the compact AuxK select compiled for occupancy 3, and the 44-vector compact instantiation replaced by a copy + the general select on
# the compact rows.  A spill that creeps into a hot kernel (round 5: a few added lines pushed fwd_fused2's slot loop over that
# threshold and every accumulator array went to the stack, 1.8 KB per lane) fails here, on the build host, instead of showing up as
# a slow GPU run.
# ---------------------------------------------------------------------------------------------------------------------
SCRATCH_ALLOWED = {}


def test_no_kernel_spills_to_scratch():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: build first")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    tmp = tempfile.mkdtemp(prefix="freud_codeobj_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        counts, cur = {}, None
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if cur and re.search(r"\bscratch_(load|store)", line):
                counts[cur] = counts.get(cur, 0) + 1
        assert len(r.stdout) > 1_000_000, "disassembly suspiciously short"
        bad = {k: v for k, v in counts.items() if v > SCRATCH_ALLOWED.get(k, 0)}
        assert not bad, f"kernels with scratch (spill) instructions: {bad}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("key", ["ff2_bf16", "ff2_bf16_pad", "ff2_f16", "ff2_f32"])
def test_fused_forward_tail_wait_counts_exactly_its_four_latent_stores(disasm, key):
    """fwd_fused2.h's epilogue overlays its staging image on the W^T ring while the loop's last LDS-DMA pieces may still be landing
    there; the hand-over is `s_waitcnt vmcnt(4)` + s_barrier: everything but the wave's FOUR youngest vector-memory operations.
    That is right only while exactly four latent stores -- and no load, no further DMA piece -- sit between the last
    global_load_lds of the kernel and that wait: an edit that adds a store or a load there would let DMA writes land
    in memory the epilogue already uses, silently.  Asserted in the built code object."""
    ins = disasm[key]
    dma = [k for k, i in enumerate(ins) if i.startswith("global_load_lds")]
    assert dma, "no LDS-DMA in the fused forward?"
    last = dma[-1]
    waits = [k for k in range(last + 1, len(ins)) if ins[k].startswith("s_waitcnt") and "vmcnt(4)" in ins[k]]
    assert waits, "the counted tail wait (vmcnt(4)) is gone: FF2_TAIL_V1's full drain, or a changed drain loop -- re-derive the count"
    w = waits[0]
    between = ins[last + 1:w]
    vm = [i for i in between if i.startswith(("global_", "buffer_", "flat_", "scratch_"))]
    stores = [i for i in vm if i.startswith("global_store")]
    assert len(stores) == 4 and len(vm) == 4, f"{len(stores)} stores / {len(vm)} vector-memory operations between the last DMA piece and the tail wait: {vm}"
    assert any(i.startswith("s_barrier") for i in ins[w:w + 4]), "the tail wait is not followed by the workgroup barrier"
