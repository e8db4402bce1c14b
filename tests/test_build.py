"""Properties of the compiled kernels that results depend on.  A render-kernel variant that spilled vector registers to
scratch once corrupted its (wave-uniform) ray-kind counters: values kept in VGPRs and spilled inside divergent code came
back stale for lanes that were inactive at the spill (rays by kind off by 30 %, nondeterministic; the image was not
affected).  Every variant must therefore fit its register budget without scratch.

(The compiler sometimes leaves a private segment of a few dead bytes behind — stack slots it reserved for scalar-register
spills and then served from vector-register lanes: no instruction touches them.  What is asserted is the property itself:
no vector-register spill, and not one scratch instruction in a render kernel.)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernel_bodies(path):
    """mangled kernel name -> the lines of its code (label to s_endpgm) in the assembly `make asm` wrote"""
    bodies, cur = {}, None
    for line in open(path):
        m = re.match(r"(_ZN2rt\w+):", line)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    return bodies


def test_no_kernel_spills_vector_registers_to_scratch():
    csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True, check=True)
    text = out.stdout + out.stderr
    kernels = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|ScratchSize \[bytes/lane\]|VGPRs|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            kernels[cur][m.group(1)] = int(m.group(2))
    trace = {k: v for k, v in kernels.items() if "trace_kernel" in k}
    assert len(trace) == 96, sorted(kernels)          # {NODE64} x {linear, BVH ring 8, BVH ring 16} x {pass 0, 1, 2, 3} x {estimator} x {counters}
    bodies = _kernel_bodies(os.path.join(csrc, "rt_hip.s"))
    for name, r in trace.items():
        assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] <= 64, (name, r)
        touching = [l for l in bodies[name] if re.match(r"\s*(scratch_|buffer_)", l)]
        assert not touching, (name, touching[:4])
        targs = name.split("trace_kernelILb")[1]
        bvh, ring16 = targs[2:5] == "Lb1", "ELi16E" in targs
        # what the launches count on: 22 waves per CU (linear scan), 15-16 (BVH, 8-entry stack ring), 12 (16-entry ring)
        assert r["Occupancy [waves/SIMD]"] >= ((3 if ring16 else 4) if bvh else 6), (name, r)
