"""CPU oracle for the deep-audio-mixer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the reported CPU baseline.

It is a plain numpy / PyTorch-CPU restatement of the reference algorithms
(each function cites the reference file:line it follows).  It is pinned
against the reference itself: ``oracle/gen_golden.py`` imports the reference
from ``/root/reference`` in the build container and writes the small fixtures
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
restatement against them.

Parity caveat: ``torchaudio.functional.amplitude_to_DB`` is a third-party
function that is absent from the image (torchaudio is not installed, the
reference pins no version).  Its documented formula is restated in
``features_ref.amplitude_to_db``; that one formula is "parity unpinned".
"""
