"""CPU oracle for the PnP-OVSS hot path: TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from this
package.  The product path (pnp-ovss_amd/) never does and fails loudly when the HIP library is
missing.  Pinning status: model/GradCAM/drop-loop/merge/upsample/blur/hist
are pinned against golden vectors produced by running the reference itself
(tests/golden/make_golden.py); the DenseCRF step is PARITY UNPINNED (pydensecrf is un-vendored and
not installable here).
"""
