"""CPU oracle for the DiffWave denoising-loop hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (PyTorch-CPU
functional ops + numpy, plus a small C file for the Cauchy kernel) of the
reference algorithm for every function on the hot path.
It exists so that the HIP engine in ``diffwave-sashimi_amd/`` can be checked
against something that does not need the reference sources at run time.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the
timed CPU baseline, never as the thing shipped.  Nothing under
``diffwave-sashimi_amd/`` imports this package; the product path fails loudly
when ``libdws.so`` is missing.

Parity pinning: every function here is pinned against golden vectors that
were produced by importing the reference modules from ``/root/reference`` in
the build container (``tests/golden/make_golden.py``, committed together with
the vectors under ``tests/golden/*.npz``), see ``tests/test_oracle_golden.py``.
The Cauchy restatement is additionally pinned by the reference's own
known-answer method (fp64 formula of ``extensions/cauchy/cauchy.py:19-26``,
``extensions/cauchy/test_cauchy.py:53-95``).

All citations are ``path:line`` relative to the reference checkout.
"""
