"""CPU oracle for the PNNP hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker.  The product path (``pnnp_amd``) never
imports this package and fails loudly when its HIP library is missing.

Contents (each function cites the reference file:line it restates):

* ``isp_np``       numpy restatement of Bayer pack/unpack (utils/isp_ops.py:57-112)
* ``noise_np``     numpy / torch-CPU restatement of the physics noise sampler
                   (data_process/process.py:591-673) driven by numpy / torch RNG
* ``net_torch``    plain torch-fp32 restatement of UNetSeeInDark / ResUnet /
                   L1 loss / PSNR / LR schedule / Adam step
* ``pnnp_oracle.c`` plain-C restatement: bit-exact pack/unpack and the
                   counter-based (Philox4x32-10) sampler specification that the
                   HIP kernel implements (tier-A parity)

Parity pinning: every function here is checked against golden vectors that
``tests/golden/make_golden.py`` produced by importing the real reference from
``/root/reference`` in the build container (see tests/test_oracle_*.py).
Third-party arithmetic behind the reference (ATen conv / RNG streams, numpy RNG
streams) is pinned to torch 2.10.0 / numpy 2.2.6 as captured in those fixtures.
"""
