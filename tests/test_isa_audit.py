"""Static audit of the generated gfx950 ISA of the kernels whose correctness leans on what hipcc does NOT know:
conv_bd_kernel's deep pipelines load their filter fragments with inline-asm buffer_load_dwordx4 and wait for them with a
hand-counted s_waitcnt vmcnt(N) — the compiler believes the destination VGPRs are valid as soon as the asm statement returns, so
a copy / spill / re-use of one of them before the wait would read data that has not landed, and no waitcnt would be inserted.
The audit compiles the file device-only to assembly (no GPU needed) and checks, for every deep-pipeline instantiation,
  * no scratch (spilled VGPRs) at all, and
  * from the first asm fragment load on, no instruction outside the asm statements — MFMAs included — reads or writes a
    fragment register while its load is still in flight (issued, not yet retired by a counted asm s_waitcnt vmcnt).
wino43_fused_kernel (LDS-DMA only, counted vmcnt) is checked for scratch and for vmcnt(0) drains inside its loop."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "treedetection_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(src, tmp_path, extra=()):
    out = str(tmp_path / (src + ".s"))
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out,
           os.path.join(CSRC, src)] + list(extra)
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return open(out).read()


def _kernels(text):
    """name → body (lines, the whole function: early exits included) of every kernel in the assembly, plus its metadata record."""
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2).splitlines()
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)(?=\n\s+- \.agpr_count|\namdhsa.target|\Z)", text, re.S):
        meta[m.group(1)] = m.group(2)
    return bodies, meta


def _regs(tok):
    """VGPR numbers named by one operand token: v12 → {12}, v[8:11] → {8..11}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _operands(line):
    code = line.split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None, []
    parts = code.split(None, 1)
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def test_conv_bd_deep_pipelines_keep_their_fragment_registers_untouched(tmp_path):
    text = _asm("conv_bdirect.hip", tmp_path)
    bodies, meta = _kernels(text)
    deep = {n: b for n, b in bodies.items() if "conv_bd_kernel" in n and re.search(r"ELi[123]ELb0EEEv8ConvArgs$", n)}
    assert len(deep) >= 9, sorted(bodies)          # 3 element-type pairs x the deep variants (the shortcut-prefetch form: next test)
    for name, body in deep.items():
        md = meta[name]
        assert re.search(r"\.vgpr_spill_count:\s+0\b", md) and re.search(r"\.private_segment_fixed_size:\s+0\b", md), name
        # Walk the k loop in text order (it is unrolled over its register sets, so text order is issue order) with the queue of
        # outstanding vector-memory operations: an asm fragment load owns its destination registers until an asm
        # s_waitcnt vmcnt(N) retires it (all but the N youngest operations — LDS-DMA loads and stores count too). While a
        # register is in flight NO instruction outside the asm statements may read or write it, MFMAs included.
        in_asm, queue, seen_load, checked = False, [], False, 0
        bad = []
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            op, ops = _operands(ln)
            if op is None:
                continue
            if in_asm:
                if op == "buffer_load_dwordx4" and ops:
                    queue.append(_regs(ops[0]))
                    seen_load = True
                m = re.search(r"vmcnt\((\d+)\)", ln)
                if op == "s_waitcnt" and m:
                    n = int(m.group(1))
                    queue = queue[len(queue) - n:] if n else []
                continue
            if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                queue.append(set())                    # LDS-DMA / stores: counted by vmcnt, no register in flight
                if op.startswith("scratch_"):
                    bad.append(ln.strip())
                continue
            if not seen_load:
                continue
            inflight = set().union(*queue) if queue else set()
            touched = set().union(*[_regs(t) for t in ops]) if ops else set()
            checked += 1
            if touched & inflight:
                bad.append(ln.strip())
        assert seen_load and checked > 100, name
        assert not bad, (name, bad[:5])


def _audit_asm_load_sites(name, body, md, min_sites):
    """Kernels whose control flow is not one straight unrolled loop (runtime-selected waits, producer / consumer roles): the local
    form of the audit. No scratch; and from every asm buffer_load_dwordx4 on, along the fall-through text until the next asm
    s_waitcnt vmcnt / branch / label, no instruction outside the asm statements reads or writes the registers the load is filling
    (what hipcc would do if it took the asm output for a finished value: a copy or a re-use right behind the statement)."""
    assert re.search(r"\.vgpr_spill_count:\s+0\b", md) and re.search(r"\.private_segment_fixed_size:\s+0\b", md), name
    assert not [ln for ln in body if ln.strip().startswith("scratch_")], name
    sites, bad = 0, []
    n = len(body)
    i = 0
    while i < n:
        if "#ASMSTART" in body[i] and i + 1 < n:
            op, ops = _operands(body[i + 1])
            if op == "buffer_load_dwordx4" and ops:
                sites += 1
                dst = _regs(ops[0])
                in_asm, j = True, i + 1
                while j + 1 < n:
                    j += 1
                    ln = body[j]
                    if "#ASMSTART" in ln:
                        in_asm = True
                        continue
                    if "#ASMEND" in ln:
                        in_asm = False
                        continue
                    st = ln.strip()
                    if re.match(r"^\.LBB\d+_\d+:", st):
                        break
                    o2, p2 = _operands(ln)
                    if o2 is None:
                        continue
                    if in_asm:
                        if o2 == "s_waitcnt" and "vmcnt" in st:
                            break
                        continue
                    if o2.startswith("s_cbranch") or o2 in ("s_branch", "s_endpgm", "s_barrier"):
                        break
                    touched = set().union(*[_regs(t) for t in p2]) if p2 else set()
                    if touched & dst:
                        bad.append(st)
        i += 1
    assert sites >= min_sites, (name, sites)
    assert not bad, (name, bad[:5])


def test_shortcut_prefetch_kernels_leave_their_in_flight_registers_alone(tmp_path):
    """conv_bd_kernel<..., RES = true> (shortcut rows behind the prologue) and conv_bs_kernel (tile id 33: the consumers' two
    shortcut register sets) load with inline asm and retire with counted waits chosen at run time."""
    bodies, meta = _kernels(_asm("conv_bdirect.hip", tmp_path))
    res = {n: b for n, b in bodies.items() if "conv_bd_kernel" in n and n.endswith("ELb1EEEv8ConvArgs")}
    assert len(res) == 1, sorted(bodies)
    for name, body in res.items():
        _audit_asm_load_sites(name, body, meta[name], 4 + 4 * 4)     # the shortcut rows + four filter register sets
    bodies, meta = _kernels(_asm("conv_bstat.hip", tmp_path))
    ks = {n: b for n, b in bodies.items() if "conv_bs_kernel" in n}
    assert len(ks) == 6, sorted(bodies)
    for name, body in ks.items():
        _audit_asm_load_sites(name, body, meta[name], 8)


def test_wino43_fused_kernel_has_no_scratch_and_no_drain_in_its_loop(tmp_path):
    text = _asm("wino_fused.hip", tmp_path)
    bodies, meta = _kernels(text)
    prod = {n: b for n, b in bodies.items() if "wino43_fused_kernel" in n and n.endswith("ELi0EEEvNS_13WinoFusedArgsE")}
    assert len(prod) == 5, sorted(bodies)          # 64-tile blocks: 8 and 4 LDS stages; 32-tile blocks (4 stages): 1, 2 and 4 ring passes per plane
    for name, body in prod.items():
        md = meta[name]
        assert re.search(r"\.vgpr_spill_count:\s+0\b", md) and re.search(r"\.private_segment_fixed_size:\s+0\b", md), name
        barriers = [i for i, ln in enumerate(body) if ln.strip().startswith("s_barrier")]
        assert len(barriers) > 20, name
        loop = body[barriers[1]:barriers[-1]]
        assert not [ln for ln in loop if "s_waitcnt" in ln and "vmcnt(0)" in ln], name      # the DMA pipeline never drains inside the loop
        assert sum("v_mfma_f32_16x16x4" in ln for ln in body) >= 16 * 6 * 4, name
