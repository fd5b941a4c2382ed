"""CPU oracle for the UNet / Swin-UNet / multi-loss / tiled-prediction hot path.

TEST INFRASTRUCTURE ONLY.  Everything under ``oracle/`` is a plain torch-CPU /
numpy fp32 restatement of the reference algorithms, written functionally over a
``state_dict``.  It exists so that the HIP path can be checked on identical
inputs.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``unet_amd`` never does.

Parity pinning: every function here is checked against the imported reference
(``tests/golden/make_golden.py``, run where ``/root/reference`` exists) and
against the frozen fixtures that script wrote into ``tests/golden/*.npz``.
"""
