"""MI355X-native (gfx950) contrastive sEMG training path.

Drop-in for the hot path of FibonacciDude/ContrastiveProsthetics: the class surface of
``code/models.py`` (``Model``), ``code/load.py`` (``DB23``), ``code/utils.py``
(``TaskWrapper``) and the ``code/train.py`` CLI, over hand-written HIP kernels behind the
C ABI of ``include/cpnative.h``.  Importing the package never touches the GPU; the first
kernel call loads ``libcpnative.so`` and fails loudly if it is absent.
"""
__version__ = "0.1.0"
