"""Import-time placeholder for the `torchtyping` annotation package (absent from this image).

Used ONLY by tests/golden/make_golden.py, in the build container, to import the reference's own
torch components; the reference uses `TensorType[...]` purely as annotations on this path.
Never imported by the product package or on the GPU box.
"""


class _Annotation:
    def __getitem__(self, item):
        return self

    def __call__(self, *args, **kwargs):
        return self


TensorType = _Annotation()


def patch_typeguard(*args, **kwargs):
    return None
