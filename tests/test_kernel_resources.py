"""Register budgets of the hot kernels (CPU: hipcc cross-compiles gfx950 and reports resource usage).  The streaming
kernels live at a chosen occupancy -- rescale+kick+drift and step_kernel at 3 work-groups per CU (<= 168 VGPRs), the
read-only kick+KE pass at 5 (<= 96) -- and nothing may spill; a harmless-looking edit (a second call site of the tile body,
a register array sized for the largest case) has cost a work-group per CU before without any test failing."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hot_kernels_keep_their_occupancy():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(\S+<[^>]*>)\s+VGPR\s+(\d+)\s+AGPR\s+\d+\s+SGPR\s+\S+\s+scratch\s+(\d+)\s+occ\s+(\d+)\s+LDS\s+(\d+)\s+vspill\s+(\d+)\s+stackops\s+(\d+)", line)
        if m:
            rows[m.group(1).replace(" ", "")] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(6)), int(m.group(7)), int(m.group(5)))
    assert len(rows) > 40, out.stdout[-2000:]
    # nothing may spill to memory: no vector-register spill, not one instruction that touches the stack (a few instantiations
    # report a 36-byte scratch size with neither: a slot reserved for spilled scalar registers that all went to VGPR lanes)
    # (chain_long_kernel<14..16> and chain_dualnh_long_kernel<12..16>, one wavefront with the chain's links in registers, move values between their vector and accumulation
    # registers -- reported as VGPR spills, but to AGPRs: its scratch and stack counts must still be zero)
    spills = {k: v for k, v in rows.items() if (v[3] != 0 and not k.startswith(("chain_long_kernel", "chain_dualnh_long_kernel"))) or v[4] != 0 or v[1] > 64}
    assert not spills, spills
    for prec in (0, 1, 2):          # the gather path (tgnh_gather.hip): untuned, but at >= 3 wavefronts per SIMD and (above) without a spill
        assert rows[f"gather_update_kernel<{prec}>"][2] >= 3 and rows[f"gather_ke_kernel<{prec}>"][2] >= 8, (prec, rows[f"gather_update_kernel<{prec}>"])
    for prec in (0, 1, 2):
        for gb in (1, 4, 8):
            for kind in range(5):        # a whole deferred step; the halves of the reference's structure, fused and split
                assert rows[f"step_kernel<{prec},{gb},{kind}>"][0] <= 168, (prec, gb, kind, rows[f"step_kernel<{prec},{gb},{kind}>"])   # 3 work-groups per CU
            assert rows[f"tile_kernel<{prec},71,1,false>"][0] <= 168 and rows[f"tile_kernel<{prec},7,1,false>"][0] <= 168      # (no KE bins: one instantiation)
            assert rows[f"tile_kernel<{prec},138,{gb},false>"][0] <= 96, rows[f"tile_kernel<{prec},138,{gb},false>"]   # 5 work-groups per CU
            # the wave-tile kernels: the KE passes at >= 5 wavefronts per SIMD, the whole-step kernel at 4 (its one-link
            # instantiation; the one for chains of 2-4 links carries the links' registers and runs at 2)
            for ops in (8, 10, 138):
                assert rows[f"wke_kernel<{prec},{ops},{gb}>"][0] <= 102, rows[f"wke_kernel<{prec},{ops},{gb}>"]
            assert rows[f"wstep_kernel<{prec},{gb},false>"][0] <= 128, rows[f"wstep_kernel<{prec},{gb},false>"]
            assert rows[f"wstep_kernel<{prec},{gb},true>"][0] <= 256, rows[f"wstep_kernel<{prec},{gb},true>"]
            # ... and the occupancies the comments in tgnh_kernels.hip, DESIGN.md and profiles/ rely on, as the compiler reports
            # them (wavefronts per SIMD): wstep_kernel 4 = two 512-thread work-groups per compute unit = 512 work-groups = 262 144
            # slots resident at once on 256 CUs, whose LDS (images + pattern constants) must fit twice into a CU's 160 KiB;
            # the KE passes at >= 5
            w = rows[f"wstep_kernel<{prec},{gb},false>"]
            assert w[2] == 4 and 2 * w[5] <= 160 * 1024, w
            assert rows[f"wstep_kernel<{prec},{gb},true>"][2] == 2
            for ops in (8, 10, 138):
                k = rows[f"wke_kernel<{prec},{ops},{gb}>"]
                assert k[2] >= 5 and k[2] * k[5] <= 160 * 1024, k      # (4 wavefronts per work-group: occupancy = work-groups per CU)
        for ops in (1, 7, 71, 65, 19):          # the rescale launches: one-link chains at 3 work-groups per CU, 2-4 links at 2
            assert rows[f"tile_kernel<{prec},{ops},1,false>"][0] <= 168, (prec, ops)
            assert rows[f"tile_kernel<{prec},{ops},1,true>"][0] <= 256, (prec, ops)
