"""Model surface of the drop-in package (see model.py)."""
from . import model as _model

SpeechDrivenAnimation = _model.SpeechDrivenAnimation
SaberSpeechDrivenAnimation = _model.SaberSpeechDrivenAnimation
__all__ = ["SpeechDrivenAnimation", "SaberSpeechDrivenAnimation"]
