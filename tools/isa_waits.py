"""s_waitcnt vmcnt(N) of every loop of every kernel in a hipcc -S listing, beside the loop's vector-memory operations per iteration.

The compiler's wait at a loop header is the minimum over the loop's own order AND the way in from the code before the loop: a kernel that
issues its first loads right before a pipelined loop (refills in place, consumed one iteration later) can end up waiting every iteration
with the small count of the entry path. This lists the candidates: loops whose waits are far below their operations per iteration.
usage: python tools/isa_waits.py build/asm/<unit>.s [kernel name filter]
  (hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o build/asm/<unit>.s densepose_torchscript_amd/csrc/<unit>.hip)"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    lines = open(sys.argv[1]).read().splitlines()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    kern, body = None, []
    kernels = []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            kern, body = m.group(1), []
            kernels.append((kern, body))
        elif kern is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                kern = None
    for name, body in kernels:
        dn = demangle(name)
        if flt and flt not in dn:
            continue
        labels = {}
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, ln in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        if not loops:
            continue
        print("==", re.sub(r"\(anonymous namespace\)::", "", dn)[:150])
        for a, b in loops:
            seg = [x.strip() for x in body[a:b + 1] if x.strip() and not x.strip().startswith(";")]
            ins = [x for x in seg if not x.startswith(".")]
            ld = sum(1 for x in ins if re.match(r"(buffer|global|flat)_load", x))
            st = sum(1 for x in ins if re.match(r"(buffer|global|flat)_store", x))
            mf = sum(1 for x in ins if x.startswith("v_mfma"))
            w = [int(m.group(1)) for x in ins for m in [re.search(r"vmcnt\((\d+)\)", x)] if m]
            if ld + st == 0 and not w:
                continue
            print("   loop lines %5d-%5d: %4d instr, %3d mfma, %2d loads, %2d stores per pass; vmcnt waits: %s" % (a, b, len(ins), mf, ld, st, w))


main()
