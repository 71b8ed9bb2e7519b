"""Host-side mirrors of the reference's module / function interfaces for the hot path.

Same class names, constructor arguments, ``forward()`` signatures, ``state_dict`` keys and error
behaviour as the reference; the computation behind them is the HIP library.
"""
