#!/usr/bin/env python3
"""In-flight-register lint for the gfx950 code objects inside libexl_amd.so.

Why: the hand-scheduled GEMM kernels issue their global loads from inline asm and count `s_waitcnt vmcnt(N)` by hand.  The
compiler does not know that such a load is still WRITING its destination registers after the asm statement: if the value is
dead in the source (the "redundant" last fetches of a straight-line loop) it may hand those registers to something else before
the wait that covers the load -- round 2's 400 x 11008 garbage (profiles/HISTORY.md 9.5) was exactly that: accumulators shuffled through
v4..v7 while a dwordx4 was still landing there.

What: disassembles every kernel, replays the vector-memory queue the way the hardware counts it on gfx9-family parts (loads,
LDS-DMA loads and stores all take a vmcnt slot and retire in issue order; `s_waitcnt vmcnt(N)` leaves the youngest N
outstanding) as a forward dataflow over the control-flow graph, and reports every instruction that reads or writes a VGPR an
outstanding load has not delivered yet, and every `s_endpgm` reached with an LDS-DMA load (`global_load_lds_*`) still outstanding
(it would land in the LDS of the next block on that CU).  Compiler-scheduled loads pass by construction (the compiler's own wait insertion uses
the same model), so every report is a hand-counting or liveness defect.

    python scripts/isa_lint.py [path/to/libexl_amd.so] [--kernel SUBSTR] [-v]
A second, unrelated rule rides along (round 4): REGISTER COPIES IN MFMA LOOPS.  With a wide accumulator (16 registers per MFMA result,
64 per wave in the prompt attention kernel) live across a branch, hipcc has kept the accumulators in two places and copied them
every iteration -- 64-96 `v_mov_b64` beside 32 MFMAs, a third of the loop's issue slots, with nothing in the source to show for it
(profiles/HISTORY.md 3, "flash_prefill8_kernel").  Every innermost loop with >= 16 MFMAs is checked for the number of registers moved by
`v_mov_b32` / `v_mov_b64` / `v_accvgpr_*` per MFMA; the library's loops sit at <= 1.4, the defect at 4-6, the limit is 2.
And a third for the one kernel that counts `lgkmcnt` by hand (flash_prefill8_kernel's K fragment reads): under every partial wait the
LDS queue may hold one kind of LDS operation only and no scalar memory read (lgkm_count_hazards).

    python scripts/isa_lint.py [path/to/libexl_amd.so] [--kernel SUBSTR] [-v]
Exit code 1 when a hazard is found.  tests/test_isa_lint.py runs it over the built library.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

VM_PREFIXES = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
               "flat_store", "flat_atomic", "scratch_load", "scratch_store")
HAND_COUNTED_PREFIXES = ("_Z15dec_ring_kernel",)   # kernels that must not spill
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(\S+)>:")


def code_objects(so_path, workdir):
    """The gfx950 code objects of every translation unit bundled into the shared library."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.devnull], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        part = os.path.join(workdir, f"bundle{i}.bin")
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, f"dev{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={part}",
                            f"--output={co}"], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernel_resources(so_path, workdir):
    """Per kernel of the library, what the code object's metadata says it occupies: {name: dict(vgpr, agpr, sgpr, scratch, lds, threads)}
    (llvm-readelf --notes: .vgpr_count, .agpr_count, .sgpr_count, .private_segment_fixed_size = scratch bytes per lane,
    .group_segment_fixed_size = static LDS, .max_flat_workgroup_size).  tests/test_kernel_resources.py holds the occupancy
    assumptions of DESIGN.md against it."""
    out = {}
    for co in code_objects(so_path, workdir):
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in txt.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            get = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)     # noqa: E731
            out[get("name")] = dict(vgpr=int(get("vgpr_count")), agpr=int(get("agpr_count")), sgpr=int(get("sgpr_count")),
                                    scratch=int(get("private_segment_fixed_size")), lds=int(get("group_segment_fixed_size")),
                                    threads=int(get("max_flat_workgroup_size")))
    return out


def disassemble(co):
    """{kernel name: [(addr, mnemonic, operands)]}"""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = _INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def vregs(text):
    s = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


_PK64 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32")
_SEL = re.compile(r"op_sel(_hi)?:\[([01,]+)\]")


def read_write_regs(mn, ops):
    """VGPRs an instruction reads or writes.  Packed-f32 instructions name 64-bit register pairs but read, per source, only the
    halves op_sel / op_sel_hi select (hipcc pairs a live scalar with whatever register sits next to it: `v[94:95]` with
    op_sel_hi 0 for that source never looks at v95)."""
    if mn not in _PK64:
        return vregs(ops)
    sel = {"": [0, 0, 0], "_hi": [1, 1, 1]}
    for m in _SEL.finditer(ops):
        vals = [int(x) for x in m.group(2).split(",")]
        sel[m.group(1) or ""] = vals + [1 if m.group(1) else 0] * (3 - len(vals))
    fields = [f.strip() for f in re.sub(r"op_sel(_hi)?:\[[01,]+\]", "", ops).split(",") if f.strip()]
    out = vregs(fields[0]) if fields else set()
    for i, f in enumerate(fields[1:4]):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", f)
        if not m:
            out |= vregs(f)
            continue
        lo = int(m.group(1))
        halves = {sel[""][i], sel["_hi"][i]}
        out |= {lo + h for h in halves}
    return out


_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def valu_sgpr_hazards(name, insns):
    """VALU write of an SGPR (v_readfirstlane / v_readlane) -> vector-memory instruction reading that SGPR (base, offset or
    descriptor) needs 5 wait states in between.  hipcc pads its own code; it cannot see into an inline-asm string, so a
    `v_readfirstlane` that lands right in front of an asm load is a silent wrong-address bug (round 3: memory access faults)."""
    out = []
    n = len(insns)
    for i, (addr, mn, ops) in enumerate(insns):
        if mn not in ("v_readfirstlane_b32", "v_readlane_b32"):
            continue
        written = sregs(ops.split(",")[0])
        states = 0
        for j in range(i + 1, min(i + 8, n)):
            a2, m2, o2 = insns[j]
            if m2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if m2.startswith(VM_PREFIXES) and states < 5:
                if written & sregs(o2):
                    out.append((name, a2, f"{m2} {o2}", addr, f"{mn} {ops} only {states} wait states earlier (5 needed)", []))
                    break
            states += (int(o2.split()[0]) + 1) if m2 == "s_nop" and o2.split() and o2.split()[0].isdigit() else 1
            if states >= 5:
                break
            if written & sregs(o2.split(",")[0]) and not m2.startswith(VM_PREFIXES) and m2.startswith("s_"):
                break                                               # the SGPR was overwritten by scalar code: a new value
    return out


def load_dest(mn, ops):
    """VGPRs a vector-memory LOAD will write when it completes (empty for stores, LDS-DMA and no-return atomics)."""
    if "_load" not in mn or "_lds" in mn or " lds" in (" " + ops):
        return set()
    first = ops.split(",")[0]
    return vregs(first)


def is_lds_dma(mn, ops):
    return "_load" in mn and ("_lds" in mn or " lds" in (" " + ops))


def branch_target(addr, mn, ops):
    if not mn.startswith(("s_cbranch", "s_branch")):
        return None
    m = re.match(r"(\d+)", ops)
    if not m:
        return None
    simm = int(m.group(1))
    if simm >= 0x8000:
        simm -= 0x10000
    return addr + 4 + 4 * simm


def vmcnt_of(ops):
    m = re.search(r"vmcnt\((\d+)\)", ops)
    return int(m.group(1)) if m else None


def lint_kernel(name, insns, verbose=False):
    """Forward dataflow over the kernel's control-flow graph.  State: {outstanding load (address): fewest vector-memory
    instructions issued after it on any path}; `s_waitcnt vmcnt(N)` retires every load with at least N younger ones; joins take
    the union with the smaller age (the path on which the load is retired latest)."""
    n = len(insns)
    if n == 0:
        return []
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    targets = {}
    leaders = {0}
    for i, (addr, mn, ops) in enumerate(insns):
        t = branch_target(addr, mn, ops)
        if t is not None and t in index:
            targets[i] = index[t]
            leaders.add(index[t])
        if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")) and i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, lo in enumerate(starts):
        hi = starts[bi + 1] if bi + 1 < len(starts) else n
        blocks.append((lo, hi))
        block_of[lo] = bi
    succ = []
    for (lo, hi) in blocks:
        addr, mn, ops = insns[hi - 1]
        out = []
        if mn == "s_endpgm" or mn.startswith("s_setpc"):
            pass
        elif mn.startswith("s_branch"):
            if hi - 1 in targets:
                out.append(block_of[targets[hi - 1]])
        else:
            if mn.startswith("s_cbranch") and hi - 1 in targets:
                out.append(block_of[targets[hi - 1]])
            if hi < n:
                out.append(block_of[hi])
        succ.append(out)
    dests = {}
    texts = {}
    for i, (addr, mn, ops) in enumerate(insns):
        if mn.startswith(VM_PREFIXES):
            d = load_dest(mn, ops)
            if d or is_lds_dma(mn, ops):                              # LDS-DMA: tracked with an empty register set (end-of-program rule)
                dests[addr] = d
                texts[addr] = f"{mn} {ops}"
    AGE_CAP = 64
    extra = []
    if name.startswith(HAND_COUNTED_PREFIXES):
        # every vector-memory instruction of these kernels is counted by hand: a register spill (scratch traffic takes vmcnt
        # slots and makes hipcc drain the queue around it) silently turns the ring into a stop-and-wait loop
        for addr, mn, ops in insns:
            if mn.startswith("scratch_"):
                extra.append((name, addr, f"{mn} {ops}", addr, "scratch traffic in a hand-counted kernel", []))
                break

    def transfer(state, lo, hi, report):
        state = dict(state)
        for i in range(lo, hi):
            addr, mn, ops = insns[i]
            if mn == "s_waitcnt":
                k = vmcnt_of(ops)
                if k is not None:
                    state = {a: g for a, g in state.items() if g < k}
                continue
            if report is not None and state and mn == "s_endpgm":
                # a DMA into LDS that outlives its block lands in the LDS of whichever block the CU runs next
                for qa in state:
                    if not dests[qa]:
                        report.setdefault((qa, addr), (name, addr, "s_endpgm", qa, texts[qa], []))
            if report is not None and state:
                # a load may target registers an older load is still writing (in-order return: the younger one wins, and the wait
                # that covers it covers the older one); only its address operands count
                touched = vregs(ops.split(",", 1)[1]) if addr in dests and "," in ops else read_write_regs(mn, ops)
                if touched:
                    for qa in state:
                        both = touched & dests[qa]
                        if both and qa != addr:
                            report.setdefault((qa, addr), (name, addr, f"{mn} {ops}", qa, texts[qa], sorted(both)))
            if mn.startswith(VM_PREFIXES):
                state = {a: min(g + 1, AGE_CAP) for a, g in state.items()}
                if addr in dests:
                    state[addr] = 0
        return state

    def merge(a, b):
        out = dict(a)
        for k, g in b.items():
            out[k] = min(out.get(k, AGE_CAP), g)
        return out

    ins = [None] * len(blocks)
    ins[0] = {}
    work = [0]
    while work:
        b = work.pop()
        out = transfer(ins[b], blocks[b][0], blocks[b][1], None)
        for s2 in succ[b]:
            new = out if ins[s2] is None else merge(ins[s2], out)
            if ins[s2] is None or new != ins[s2]:
                ins[s2] = new
                work.append(s2)
    report = {}
    for b, (lo, hi) in enumerate(blocks):
        if ins[b] is not None:
            transfer(ins[b], lo, hi, report)
    return list(report.values()) + extra + valu_sgpr_hazards(name, insns)


HAND_COUNTED_LGKM_PREFIXES = ("_Z21flash_prefill8_kernel",)   # kernels with hand-issued LDS reads under counted lgkmcnt waits


def lgkm_count_hazards(name, insns):
    """A partial wait `s_waitcnt lgkmcnt(N > 0)` retires the OLDEST LDS operations (they return in order); a scalar memory read in the
    queue returns out of order and makes the count meaningless, and a hand-counted sequence only holds while nothing else enters the queue
    between its first read and its last wait.  In the kernels named above every partial wait must therefore see, since the last full
    drain of its basic block, LDS operations of ONE kind only and no scalar memory operation: an LDS read of another kind that the
    compiler or the scheduler lifted into a counted stretch shows up as a second kind."""
    if not name.startswith(HAND_COUNTED_LGKM_PREFIXES):
        return []
    out, queue = [], []
    for addr, mn, ops in insns:
        if mn.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
            queue = []                                                 # (every block of these kernels is entered behind a full drain or a barrier)
            continue
        if mn.startswith(("ds_", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            queue.append((addr, mn))
            continue
        if mn == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ops)
            if m is None:
                continue
            n = int(m.group(1))
            if n > 0:
                kinds = sorted({q for _, q in queue})
                scalar = [q for q in kinds if q.startswith("s_")]
                if scalar or len(kinds) > 1:
                    out.append((name, addr, f"s_waitcnt {ops}", queue[0][0] if queue else addr,
                                "partial LDS wait with " + (", ".join(kinds)) + " in the queue: a counted wait needs one kind of LDS operation and no scalar read", []))
            queue = queue[len(queue) - n:] if n and n < len(queue) else ([] if n == 0 else queue)
    return out


COPY_RULE_MIN_MFMA = 16
COPY_RULE_LIMIT = 2.0


def accumulator_copy_hazards(name, insns):
    """Innermost loops with >= COPY_RULE_MIN_MFMA MFMAs that move >= COPY_RULE_LIMIT registers per MFMA with plain copies."""
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    loops = []
    for i, (addr, mn, ops) in enumerate(insns):
        t = branch_target(addr, mn, ops)
        if t is not None and t in index and index[t] <= i:
            loops.append((index[t], i))
    out = []
    for lo, hi in loops:
        if any(m != (lo, hi) and m[0] >= lo and m[1] <= hi for m in loops):
            continue                                                   # not innermost
        body = insns[lo:hi + 1]
        mfma = sum(1 for _, mn, _ in body if "mfma" in mn)
        if mfma < COPY_RULE_MIN_MFMA:
            continue
        regs = sum(2 if mn.startswith("v_mov_b64") else 1 for _, mn, _ in body if mn.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_")))
        if regs >= COPY_RULE_LIMIT * mfma:
            out.append((name, insns[lo][0], f"loop of {hi - lo + 1} instructions with {mfma} MFMAs", insns[lo][0],
                        f"{regs} registers copied per iteration (v_mov / v_accvgpr): accumulators kept in two places?", []))
    return out


def lint_library(so_path, kernel_filter=None, verbose=False):
    """-> (number of kernels checked, [hazard tuples])"""
    hazards, count = [], 0
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(so_path, wd):
            for name, insns in disassemble(co).items():
                if kernel_filter and kernel_filter not in name:
                    continue
                if not insns:
                    continue
                count += 1
                hz = lint_kernel(name, insns, verbose) + accumulator_copy_hazards(name, insns) + lgkm_count_hazards(name, insns)
                if verbose:
                    print(f"{name[:100]}: {len(insns)} instructions, {len(hz)} hazards")
                hazards += hz
    return count, hazards


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("-")]
    verbose = "-v" in argv
    kf = None
    if "--kernel" in argv:
        kf = argv[argv.index("--kernel") + 1]
        args = [a for a in args if a != kf]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exllama_amd", "libexl_amd.so")
    count, hazards = lint_library(so, kf, verbose)
    for (name, addr, text, qa, qtext, regs) in hazards:
        if regs:
            print(f"HAZARD {name[:90]}\n   {addr:#x}: {text}\n   touches v{regs} while the load at {qa:#x} is outstanding: {qtext}")
        else:
            print(f"HAZARD {name[:90]}\n   {addr:#x}: {text}\n   {qtext}")
    print(f"{count} kernels checked, {len(hazards)} hazards")
    return 1 if hazards else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
