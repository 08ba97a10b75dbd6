"""Mirror of the reference's SBL_Multilingual_Lip_reading/config.py (module-level constants imported by name
from train.py:11, test.py:11, decoder.py:8, loss.py:4).  Same names, same values; dataset paths are kept so the
reference's scripts import unchanged."""
import torch

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')  # config.py:5

# Model parameters (audio leftovers of the Speech-Transformer fork, unused on the hot path: config.py:8-17)
input_dim = 80
window_size = 25
stride = 10
hidden_size = 512
embedding_dim = 512
cmvn = True
num_layers = 4
LFR_m = 4
LFR_n = 3
sample_rate = 16000

# Training parameters (config.py:20-22)
grad_clip = 5.
print_freq = 50
checkpoint = None

# Data parameters (config.py:25-38)
IGNORE_ID = -1
sos_id = 0
eos_id = 1
num_train = 120098
num_dev = 14326
num_test = 7176
word_number = 6
p = 1
mask = 0.7
vocab_size = 58

lrw_path = '../roi_80_116_175_211_npy_gray'
lrw_info = '../LRW_TXT'
lrw_wav = '../lrw_mp4'
lrw1000_path = '../LRW1000_npy_rsz122_gray'
lrw1000_info = '../LRW1000_info'
lrw1000_wav = '../LRW1000_audio'

# Values the reference hard-codes inside the model ("config / flags"), exposed as constants
MAX_DECODE_LEN = 16            # decoder.py:95,303; transformer/utils.py:5
TEACHER_COIN_THRESHOLD = 0.5   # decoder.py:176
FRONTEND_DROPOUT_P = 0.5       # video_frontend.py:122 (active in eval too)
