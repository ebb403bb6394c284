class _Placeholder:       # names only; see the package docstring
    def __init__(self, *a, **k):
        raise NotImplementedError("placeholder")


class Resize(_Placeholder): pass
class Compose(_Placeholder): pass
class ToTensor(_Placeholder): pass
class Normalize(_Placeholder): pass
class ColorJitter(_Placeholder): pass
class RandomHorizontalFlip(_Placeholder): pass
class RandomResizedCrop(_Placeholder): pass
