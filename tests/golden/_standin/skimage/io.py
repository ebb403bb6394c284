def imread(*a, **k):      # name only; see the package docstring
    raise NotImplementedError("placeholder")
