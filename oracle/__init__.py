"""CPU oracle for the YoloLite inference hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``yololite-official-repo_amd/`` imports
it; the product path fails loudly when the HIP extension is missing instead of
falling back to this code.

What is restated here (plain PyTorch-CPU fp32 / numpy), and where it comes from:

* ``oracle.backbones``   timm feature extractors the reference obtains through
  ``timm.create_model(..., features_only=True)`` (reference call sites
  ``scripts/model/model_v2.py:94,98-100,266,270-272``).  ``timm`` is a third-party
  dependency pinned only as ``timm>=0.9`` (``requirements.txt:3``) and is NOT present
  in ``/root/reference`` nor in this image.  The restatement follows timm's published
  MobileNetV4 / EfficientNet-Lite definitions from recollection with a parameter-count
  checksum (0.5524 M params for edge_n at C=3 vs. 0.553 M published,
  ``BENCHMARK.md:353``).  **Parity unpinned** for the backbone.
* ``oracle.model``       FPN neck + decoupled heads + output layout
  (``scripts/model/model_v2.py:15-64,77-247,250-383``).  Pinned against the reference's
  own classes imported under stubs (``tests/golden/make_fixtures.py``).
* ``oracle.postproc``    anchor-free decode (``scripts/helpers/utils_ms.py:26-123``),
  the three score/threshold/NMS pipelines (``tools/infer.py:460-493`` main,
  ``tools/infer.py:247-389`` fallback, ``scripts/helpers/helpers.py:87-153`` eval),
  back-mapping (``tools/infer.py:508-516``).  Decode, the fallback pipeline (with the
  reference's own pure-torch greedy NMS) and the control flow of the main/eval
  pipelines are pinned against the reference under stubs.  ``torchvision.ops.nms``
  itself (``torchvision>=0.17``, absent here) is restated from its published CPU
  kernel: **parity unpinned** for that primitive, cross-checked against the
  reference's fallback NMS on tie-free inputs.
* masks: the reference contains no mask/proto code at all -> **parity unpinned**.
"""
