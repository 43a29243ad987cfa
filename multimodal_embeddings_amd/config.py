"""Constants of the hot path, mirrored from deprecated_package/config.py (file:line cited)."""

MAX_IMAGE_HEIGHT_AND_WIDTH = 8000  # config.py:18  (embedder.py:110-114 LANCZOS cap)
BATCH_SIZE = 16  # config.py:51  (only shapes work lists in the reference, G9)
REGION_BATCH_SIZE = 48  # config.py:52
REGION_TYPES_TO_PROCESS = [  # config.py:67-74
    "title",
    "plain_text",
    "figure",
    "figure_caption",
    "table",
    "table_caption",
]
REGION_COMPARE_TOP_N = 10  # config.py:77
REGION_SIMILARITY_THRESHOLD = 0.3  # config.py:78 (ignored by wrc:151, see G5)
CROSS_COMPARE_TOP_N = 5  # config.py:22
EFFECTIVE_THRESHOLD = 0.1  # wrc:151 hard-coded
PAGE_QUERY_REGIONS = 10  # wrc:199
PAGE_TOP_K = 10  # wrc:210
PREFIX_LENGTH = 20  # wrc:97 default

DEFAULT_MODEL_NAME = "synthetic/vit-b16-224-seed1"  # stands in for config.py:58 (network fetch unavailable)
# CLIP statistics shipped by the real checkpoint's preprocessor_config
IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)
EMBED_DIM = 768
WEIGHT_BY_AREA = True  # config.py:79
