"""MI355X-native drop-in for the reference's `tt` package (model / encoder / decoder / transformer /
utils keep their class names, constructor signatures, parameter names and forward contracts); the
arithmetic runs in libttmi's HIP kernels through `ttmi.ops`."""
