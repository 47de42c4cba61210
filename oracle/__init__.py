"""CPU oracle for the SNVC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  Nothing under
``snvc_amd/`` imports it; the product path raises if its HIP library is absent.

Contents
  * ``oracle.native``     -- ctypes bindings of the C restatements
                             (``cost_volume_ref.c``, ``roiaware_pool3d_ref.c``).
  * ``oracle.torch_ref``  -- PyTorch-CPU restatement of the reference's 3D module
                             graphs and of ``_sample_2d_feat`` (what the reference
                             itself executes on CPU: plain ``torch.nn``).
  * ``oracle.numpy_ref``  -- numpy restatements of small index/gather routines.
"""
