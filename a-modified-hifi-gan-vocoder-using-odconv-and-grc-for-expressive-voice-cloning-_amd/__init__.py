"""MI355X-native conditioned HiFi-GAN vocoder path (package root).

The directory name mirrors the reference repository and is not a Python identifier; put this
directory on ``sys.path`` and import the drop-in package ``hifigan_modified`` (same module and class
names as the reference's ``hifigan_modified``), or load it with ``importlib``.
"""
