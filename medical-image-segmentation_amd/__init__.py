"""MI355X-native segmentation-network forward pass behind the reference's
`utils/model.py` surface.  The directory name is not a Python identifier;
import it as `miseg_amd` (alias module at the repository root) or with
`importlib.import_module("medical-image-segmentation_amd")`.
"""
__version__ = "0.1.0"
