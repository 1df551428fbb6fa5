from .sliding_window import sliding_window, sliding_window_inference

__all__ = ['sliding_window', 'sliding_window_inference']
