"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch-CPU fp32 / numpy) of the reference hot path of zhouxian/act3d-chained-diffuser.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
package `act3d-chained-diffuser_amd/` never does and fails loudly without its HIP library.

Pinning: the reference publishes no tests or golden vectors for this path.  The oracle is pinned
against outputs of the reference itself, generated in the build container by tests/golden/make_goldens.py
(which imports /root/reference with third-party stubs) and committed as tests/golden/*.pt; see
tests/test_oracle_golden.py.  Third-party arithmetic that is absent from /root/reference -- diffusers'
DDPMScheduler (un-pinned version), torchvision's FeaturePyramidNetwork, openai-CLIP RN50 -- is restated from
its published algorithm and is PARITY UNPINNED at that boundary.
"""
