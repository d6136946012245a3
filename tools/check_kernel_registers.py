#!/usr/bin/env python3
"""Build-time guard for the inline-asm kernels (linear_wide_kernel.h, linear_chain_kernel.h, wgrad_wide.hip, wgrad_x9.hip).

Their operand loads are issued by inline asm long before use, so the compiler does not know that those registers are "in
flight": a register copy or spill it inserts between such a load and the s_waitcnt that lands it reads garbage (seen once:
wrong Y2 rows in the first 8 rows of every tile when deferred stores pushed the CHAIN kernel over 256 VGPRs).  This tool
disassembles each kernel and scans it linearly:
  * a VGPR becomes in flight when a global_load_* / ds_read_* writes it and lands at the s_waitcnt that covers it: vector-memory
    loads retire in order, so vmcnt(N) lands all but the N newest (wgrad_x9_kernel keeps a rolling window of loads in flight);
    LDS reads land at lgkmcnt(0) only - partial counts are ignored there, which only makes the check stricter;
  * v_mov_b32 / v_accvgpr_write_b32 / scratch_store / v_writelane reading an in-flight VGPR is a violation;
  * any scratch usage is a violation;
  * a vector-memory instruction that reads an SGPR (descriptor, offset, base) which v_readlane / v_readfirstlane wrote fewer
    than 5 wait states earlier is a violation (seen as GPU memory faults, twice);
  * a vector-ALU instruction that reads the result of an MFMA less than 18 wait states (s_nop) after it issued, with no other
    MFMA in between, is a violation: linear_wide_kernel's MFMAs are inline asm, which the compiler's hazard recogniser does
    not see (seen once: register copies of the accumulators placed right behind a tile's last MFMA gave wrong values in the
    rows that MFMA writes last).
  * linear_wide_kernel keeps its weight slab in AGPRs that are still being loaded while the first tile runs: ANY v_accvgpr_*
    instruction there is a violation (seen once: a wait statement naming one AGPR quadruple twice made the compiler rotate
    the slab through copies).
Usage: check_kernel_registers.py <hipcc> <csrc dir> [file.hip ...]   (exit code 1 on violation)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

KERNELS = re.compile(r"linear_wide_kernel|linear_chain_kernel|wgrad_wide_kernel|wgrad_x9_kernel")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(code, no_agpr_moves=False):
    inflight_g, inflight_l, bad = set(), set(), []
    gqueue = []                         # destination registers of the vector-memory loads in flight, oldest first
    mfma_dst, mfma_wait = set(), 0      # result registers of the most recent MFMA and the wait states seen since it issued
    valu_sgpr = {}                      # SGPR written by the vector ALU (v_readlane / v_readfirstlane) -> wait states since
    for n, line in enumerate(code):
        s = line.split(";")[0].strip()
        if not s or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if no_agpr_moves and op.startswith("v_accvgpr"):
            bad.append((n, s + "   [copy of a weight-slab AGPR]", []))
        # a scalar operand written by the vector ALU needs 5 wait states before a vector-memory instruction reads it; the compiler
        # inserts them for its own instructions, not for inline asm (every asm VMEM statement here carries its own s_nop 4)
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            for k in list(valu_sgpr):
                valu_sgpr[k] += 1
            valu_sgpr.update({r: 0 for r in sregs(ops[0])})
        else:
            step = int(rest.strip(), 0) + 1 if op == "s_nop" else 1
            if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_"):
                used = set()
                for o in ops:
                    used |= sregs(o.split()[0])
                hit = sorted(r for r in used if r in valu_sgpr and valu_sgpr[r] < 5)
                if hit:
                    bad.append((n, s + "   [scalar operand written by the vector ALU %d wait states earlier]" % min(valu_sgpr[r] for r in hit), hit))
            for k in list(valu_sgpr):
                valu_sgpr[k] += step
                if valu_sgpr[k] >= 5:
                    del valu_sgpr[k]
        # inline-asm MFMAs are invisible to the hazard recogniser: a vector-ALU read of a 16-pass MFMA's result needs 18 wait
        # states; another MFMA in between occupies the pipe for 64 cycles, s_nop N counts N + 1
        if op.startswith("v_mfma"):
            mfma_dst, mfma_wait = regs(ops[0]), 0
            continue
        if op == "s_nop":
            mfma_wait += int(rest.strip(), 0) + 1
            continue
        if mfma_dst and mfma_wait < 18 and op.startswith("v_"):
            srcs = set()
            for o in ops[1:]:
                srcs |= regs(o)
            hit = srcs & mfma_dst
            if hit:
                bad.append((n, s + "   [reads an MFMA result %d wait states after it issued]" % mfma_wait, sorted(hit)))
        if mfma_dst and not op.startswith("s_") and not op.startswith("ds_") and not op.startswith("buffer_") and not op.startswith("global_"):
            mfma_wait += 1
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                # vector-memory loads retire in order: vmcnt(N) lands all but the N newest (stores also count, which only makes this
                # stricter: they are not in the queue here, so fewer loads are taken to have landed than really have)
                keep = int(m.group(1))
                gqueue[:] = gqueue[len(gqueue) - keep:] if keep else []
                inflight_g.clear()
                for d in gqueue:
                    inflight_g.update(d)
            if "lgkmcnt(0)" in rest:
                inflight_l.clear()
            continue
        if op.startswith("global_load") or op.startswith("buffer_load"):
            inflight_g |= regs(ops[0])
            gqueue.append(regs(ops[0]))
            continue
        if op.startswith("ds_read"):
            inflight_l |= regs(ops[0])
            continue
        if op in ("v_mov_b32_e32", "v_mov_b32", "v_accvgpr_write_b32", "v_writelane_b32") or op.startswith("scratch_store"):
            srcs = set()
            for o in ops[1:]:
                srcs |= regs(o)
            if op.startswith("scratch_store"):
                srcs = set().union(*[regs(o) for o in ops])
            hit = srcs & (inflight_g | inflight_l)
            if hit:
                bad.append((n, s, sorted(hit)))
        # a register overwritten by ordinary code is no longer "in flight" for our purposes
        if op.startswith("v_") and ops and not op.startswith("v_mfma") and not op.startswith("v_cmp"):
            d = regs(ops[0])
            inflight_g -= d
            inflight_l -= d
            for q in gqueue:
                q -= d
    return bad


def check(hipcc, src, inc):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + inc, "-S", "--cuda-device-only", "-w", src, "-o", out], check=True)
        t = open(out).read()
    res = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
        name, body = m.group(1), m.group(2)
        if not KERNELS.search(name):
            continue
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\S+)", body).group(1))
        fm = re.search(re.escape(name) + r":.*?s_endpgm", t, re.S)
        bad = scan(fm.group(0).split("\n"), no_agpr_moves=("linear_wide_kernel" in name or "linear_chain_kernel" in name))
        if scratch != 0 or bad:
            res.append((name, scratch, bad))
    return src, res


def main():
    hipcc, csrc = sys.argv[1], sys.argv[2]
    files = sys.argv[3:] or [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))
                             if re.match(r"(linear_wide_inst_|linear_chain_inst_|wgrad_wide|wgrad_x9).*\.hip$", f)]
    inc = os.path.join(csrc, "..", "..", "include")
    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(lambda f: check(hipcc, f, inc), files))
    nbad = 0
    for src, res in results:
        for name, scratch, bad in res:
            nbad += 1
            print(f"{os.path.basename(src)}: {name}: scratch {scratch} B, {len(bad)} copies of in-flight registers")
            for n, s, hit in bad[:5]:
                print(f"    line {n}: {s}   (in flight: v{hit})")
    print("checked", len(files), "files:", "OK" if nbad == 0 else f"{nbad} kernels touch registers whose loads are in flight")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
