"""Anchor constants of the reference (config.py:33-37)."""
import typing


class AnchorsConfig(typing.NamedTuple):
    sizes: typing.Sequence[int] = (32, 64, 128, 256, 512)
    strides: typing.Sequence[int] = (8, 16, 32, 64, 128)
    ratios: typing.Sequence[float] = (1, 2, .5)
    # present in the reference, never read: AnchorGenerator hard-codes its scales (anchors.py:37-40)
    scales: typing.Sequence[float] = (2 ** 0, 2 ** (1 / 3.0), 2 ** (2 / 3.0))
