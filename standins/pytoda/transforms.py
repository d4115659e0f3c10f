"""`pytoda.transforms.Compose` (imported by the reference's utils/utils.py:13):
apply callables in order."""


class Compose:

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for transform in self.transforms:
            sample = transform(sample)
        return sample
