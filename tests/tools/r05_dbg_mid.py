import os, sys, warnings
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch
import hpc_torch_utils_network as NW
from hpc_rll.torch_utils.network.rnn import LSTM
dev = torch.device("cuda:0")
for (S,B,I,H) in [(6,16,32,1024),(6,16,32,512),(4,64,32,1024)]:
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, 1, check_persistent=True).to(dev)
    x = torch.randn(S, B, I, device=dev)
    with torch.no_grad():
        ref, _ = m(x, None)
    torch.cuda.synchronize()
    print((S,B,I,H), "path", NW.lstm_last_forward_path(), "async", NW.async_error())
