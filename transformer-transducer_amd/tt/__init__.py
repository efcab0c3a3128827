"""MI355X-native drop-in for the reference's `tt` package (model / encoder / decoder / transformer /
utils keep their class names, constructor signatures, parameter names and forward contracts); the
arithmetic runs in libttmi's HIP kernels through `ttmi.ops`.

Overlay semantics: put this directory BEFORE the reference checkout on sys.path.  The hot-path modules
(`tt.model`, `tt.encoder`, `tt.decoder`, `tt.transformer`, and the hot-path names of `tt.utils`) resolve to
this package; every other `tt.*` module the training scripts import (`tt.dataset`, `tt.optim`,
`tt.kaldi_io`, ...) and every other `tt.utils` helper (logging, checkpoint, feature extraction) resolves to the
reference's own files further down the path - nothing outside the accelerated path is re-implemented here.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)      # later sys.path entries that also hold a `tt/` directory are searched too
