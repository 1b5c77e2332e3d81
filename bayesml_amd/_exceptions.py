"""Error and warning classes of the drop-in boundary.

Same names and meaning as the reference's ``bayesml/_exceptions.py:3-25`` so that user code
catching them keeps working: bad constants / hyper-parameters -> ParameterFormatError, bad data ->
DataFormatError, unknown loss -> CriteriaError; ResultWarning / ParameterFormatWarning are
UserWarning subclasses.
"""


class _ValueCarrier(Exception):
    """Keeps the offending message in ``.value`` and prints its repr, like the reference's errors."""

    def __init__(self, value):
        super().__init__(value)
        self.value = value

    def __str__(self):
        return repr(self.value)


class ParameterFormatError(_ValueCarrier):
    pass


class DataFormatError(_ValueCarrier):
    pass


class CriteriaError(_ValueCarrier):
    pass


class ResultWarning(UserWarning):
    pass


class ParameterFormatWarning(UserWarning):
    pass
