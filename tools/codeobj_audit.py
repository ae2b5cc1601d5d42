#!/usr/bin/env python3
"""What the compiler made of the kernels: a per-kernel table from the SHIPPED code object (the gfx950 bundle inside
libconsent_amd.so): registers, spills, scratch bytes per lane, static counts of scratch_* / global / LDS / DPP / packed instructions.
Runs in the GPU-less container (llvm-readelf --notes + llvm-objdump -d on the unbundled object).

  python tools/codeobj_audit.py [path/to/libconsent_amd.so] [--all] [--loops KERNEL]

--all: every kernel (default: the kernels of a step and of the driver's pipeline); --loops K: where the scratch accesses of
kernel K (substring of its name) sit, as (basic block label, instruction) lines.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unbundle(lib, tmp):
    fat = os.path.join(tmp, "fatbin")
    co = os.path.join(tmp, "gfx950.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}", "--unbundle"])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    short = []
    for d in out:
        d = re.sub(r"\(.*$", "", d)  # drop the argument list
        d = re.sub(r"^void ", "", d)
        short.append(d)
    return dict(zip(names, short))


def notes(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    kernels, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "args":
            continue
        if key == "group_segment_fixed_size" and "name" in cur:
            kernels.append(cur)
            cur = {}
        if key in ("name", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
                   "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size", "uses_dynamic_stack"):
            if key == "name" and "name" in cur:
                continue  # argument names come before the kernel's own .name inside .args; the kernel's is the last one seen
            cur[key] = val
    if "name" in cur:
        kernels.append(cur)
    return [k for k in kernels if "vgpr_count" in k]


def disasm(co):
    """{symbol: [(label, mnemonic, text)]} with, per instruction, the innermost loop it sits in: labels are llvm-objdump's
    --symbolize-operands block labels (L0, L1, ... per function); a loop is the span between a label and a later branch back to it."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", co], capture_output=True, text=True).stdout
    fns, cur, label = {}, None, ""
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            sym = m.group(1)
            if re.fullmatch(r"L\d+", sym):
                label = sym
                fns[cur].append((label, "", ""))  # a label marker (mnemonic empty)
            else:
                cur, label = sym, ""
                fns[cur] = []
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith("//"):
            continue
        s = re.sub(r"\s*//.*$", "", s)
        mn = s.split()[0] if s.split() else ""
        if mn:
            fns[cur].append((label, mn, s))
    return {k: [x for x in v] for k, v in fns.items()}


def loops_of(ins):
    """[(first index, last index)] of the loops of one function: a branch at index j to a label placed at index i <= j."""
    where = {}
    for i, (label, mn, s) in enumerate(ins):
        if mn == "":
            where[label] = i
    out = []
    for j, (label, mn, s) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = s.split()[-1]
            if tgt in where and where[tgt] <= j:
                out.append((where[tgt], j))
    return out


def innermost(loops, i):
    best = None
    for a, b in loops:
        if a <= i <= b and (best is None or b - a < best[1] - best[0]):
            best = (a, b)
    return best


def count(ins, pred):
    return sum(1 for _, mn, s in ins if mn and pred(mn, s))


STEP = ("cw_index_kernel", "cw_chain_kernel", "cw_poa_kernel", "cw_poa_q_kernel", "cw_poa_slab_kernel", "cw_poa_big_kernel", "cw_finish_kernel",
        "cw_stitch_kernel", "cw_extract", "cw_sort_tier_kernel", "cw_setup_kernel")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "consent_amd", "libconsent_amd.so")
    every = "--all" in sys.argv
    loops = sys.argv[sys.argv.index("--loops") + 1] if "--loops" in sys.argv else None
    if loops and loops in args:
        args.remove(loops)
        lib = args[0] if args else os.path.join(ROOT, "consent_amd", "libconsent_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        co = lib if lib.endswith(".co") else unbundle(lib, tmp)  # a device-only object (hipcc --cuda-device-only -c) is read as it is
        ks = notes(co)
        fns = disasm(co)
    names = demangle([k["name"] for k in ks])
    print(f"# {os.path.relpath(lib, ROOT)}: gfx950 code object, {len(ks)} kernels (llvm-readelf --notes, llvm-objdump -d)")
    print("# scratch = private_segment_fixed_size, bytes per lane; waves/SIMD = what the VGPR count allows (512 / VGPRs, at most 8)")
    hdr = f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'LDS_B':>7s} {'w/SIMD':>6s} {'insts':>7s} {'scr_ld':>6s} {'scr_st':>6s} {'glob':>5s} {'ds':>5s} {'dpp':>5s} {'v_pk':>5s} {'scr_in_loop':>11s}"
    print(hdr)
    for k in ks:
        nm = names[k["name"]]
        if not every and not any(s in nm for s in STEP):
            continue
        ins = fns.get(k["name"], [])
        lp = loops_of(ins)
        in_loop = sum(1 for i, (_, mn, _s) in enumerate(ins) if mn.startswith("scratch_") and innermost(lp, i))
        v = int(k["vgpr_count"]) + int(k.get("agpr_count", 0) or 0)
        gran = max(8, (v + 7) // 8 * 8)
        wps = min(8, 512 // gran)
        print(f"{nm[:58]:58s} {k['vgpr_count']:>5s} {k['sgpr_count']:>5s} {k['vgpr_spill_count']:>6s} {k['sgpr_spill_count']:>6s} "
              f"{k['private_segment_fixed_size']:>8s} {k['group_segment_fixed_size']:>7s} {wps:>6d} {sum(1 for x in ins if x[1]):>7d} "
              f"{count(ins, lambda m, s: m.startswith('scratch_load')):>6d} {count(ins, lambda m, s: m.startswith('scratch_store')):>6d} "
              f"{count(ins, lambda m, s: m.startswith('global_') or m.startswith('flat_') or m.startswith('buffer_')):>5d} "
              f"{count(ins, lambda m, s: m.startswith('ds_')):>5d} {count(ins, lambda m, s: 'dpp' in s or 'row_shr' in s or 'row_bcast' in s):>5d} "
              f"{count(ins, lambda m, s: m.startswith('v_pk_')):>5d} {in_loop:>11d}")
    if loops:
        for k in ks:
            nm = names[k["name"]]
            if loops not in nm:
                continue
            print(f"\n# scratch accesses of {nm}: block label, innermost enclosing loop (its label, instructions in it) or '-', instruction")
            ins = fns.get(k["name"], [])
            lp = loops_of(ins)
            for i, (label, mn, s) in enumerate(ins):
                if mn.startswith("scratch_"):
                    lo = innermost(lp, i)
                    inner = f"{ins[lo[0]][0]}:{sum(1 for x in ins[lo[0]:lo[1] + 1] if x[1])}" if lo else "-"
                    print(f"  {label:8s} {inner:12s} {s}")


if __name__ == "__main__":
    main()
