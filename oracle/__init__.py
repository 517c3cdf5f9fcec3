"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the gate-application hot path.

Nothing under ``oracle/`` is part of the shipped engine.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU column.  The product
package ``quantum_simulations_amd`` never imports this directory and has no
CPU fallback: without the HIP library it raises.
"""
