"""CPU oracle for the nind-denoise tiled-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker / the timed CPU baseline -- never as a fallback for the HIP path.

Contents
  tiler.py     numpy restatement of ``OneImageDS`` + the crop->stitch loop of
               ``denoise_image.py`` (integer / byte-exact work).
  networks.py  torch.nn.functional (CPU, fp32) restatement of ``UtNet.forward``
               and ``UNet.forward`` driven by a plain state-dict.
  cref/        plain-C restatement of the tile geometry, gather, stitch and a
               naive direct convolution, built by ``oracle/cref/Makefile``.

Pinning: the reference holds no golden vectors or known-answer tests for this
path.  The oracle is pinned by fixtures under
``tests/golden/`` produced by importing the reference's own ``UtNet`` /
``OneImageDS`` in the build container (``tests/golden/make_golden.py``).
"""
