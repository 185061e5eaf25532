def polyblur_deblurring(*a, **k):
    raise NotImplementedError
class PolyblurDeblurring:
    pass
