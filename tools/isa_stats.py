"""Register / spill statistics of the elimination kernel's instantiations from the ISA (no GPU needed):
    python tools/isa_stats.py [extra hipcc flags ...]
compiles rlap_amd/csrc/rlap_kernels.hip for gfx950 to assembly (device only, about three minutes; once per translation unit of
csrc/Makefile) and prints, per instantiation of
k_eliminate_batch_t, the instruction count, SGPRs / VGPRs, scratch size and the number of scratch loads / stores and of
v_readlane / v_writelane (SGPR spills parked in VGPR lanes).  Why: the kernel sits at both register limits, and DESIGN.md section 5
(round 3) found that its run time follows where the allocator puts its reloads -- with 79 scratch loads in <degree, asc, 32, 1024>
C3 ran in 283 ms, every build with 130-160 of them in about 300, the present one (12, machine LICM off for the priority-queue
kernels) in 278.  A change can be screened here before it is measured."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    # the two translation units of csrc/Makefile: everything but the priority-queue instantiations, then those (machine LICM off)
    for tag, extra in (("rlap_kernels.o", []), ("rlap_kernels_pq.o", ["-DRLAP_ELIM_PQ_TU", "-mllvm", "-disable-machine-licm"])):
        print(f"--- {tag}: hipcc ... {' '.join(extra + sys.argv[1:])}")
        table(extra + sys.argv[1:])


def table(flags):
    out = os.path.join(tempfile.gettempdir(), "rlap_kernels_isa.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *flags,
           "-S", "--cuda-device-only", os.path.join(ROOT, "rlap_amd", "csrc", "rlap_kernels.hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    print(f"{'instantiation <o_v,o_n,slots,threads>':40s} {'instr':>6s} {'sgpr':>5s} {'vgpr':>5s} {'scratchB':>8s} {'s_load':>6s} {'s_store':>7s} {'readlane':>8s} {'writelane':>9s}")
    for i, l in enumerate(lines):
        m = re.match(r"^_ZN4rlap19k_eliminate_batch_tILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\w*:", l)
        if not m:
            continue
        end = i
        while not lines[end].startswith(".Lfunc_end"):
            end += 1
        body = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
        c = collections.Counter(x.split()[0] for x in body if x)
        meta = "\n".join(lines[end:end + 300])
        g = lambda k: (re.search(k + r":\s*(\d+)", meta) or [None, "?"])[1]
        name = "<%s,%s,%s,%s>" % m.groups()
        print(f"{name:40s} {len(body):6d} {g('TotalNumSgprs'):>5s} {g('NumVgprs'):>5s} {g('ScratchSize'):>8s} "
              f"{sum(v for k, v in c.items() if k.startswith('scratch_load')):6d} {sum(v for k, v in c.items() if k.startswith('scratch_store')):7d} "
              f"{c['v_readlane_b32']:8d} {c['v_writelane_b32']:9d}")


if __name__ == "__main__":
    main()
