from .model import SaberSpeechDrivenAnimation, SpeechDrivenAnimation  # noqa: F401
