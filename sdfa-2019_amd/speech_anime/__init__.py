"""Drop-in host surface for the inference path of chaiyujin/sdfa-2019, backed by libsdfa_hip.so (MI355X).

Keeps the reference's module / class / method names for the hot path so that
`python3 -m speech_anime evaluate ...` (evaluate.sh:15-22) and callers of
`SaberSpeechDrivenAnimation.generate_animation`, `DatasetSlidingWindow.fetch_audio_features` and
`SpeechDrivenAnimation.forward` keep working; everything arithmetic runs in the HIP library
(include/sdfa_hip.h).  Training, rendering and the mesh solve are out of scope.
"""
from . import api  # noqa: F401
