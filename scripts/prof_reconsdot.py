"""Stage timing of the reconstruction distance (openpvsg_amd/unitrack.py reconsdot_cost) on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbench import timeit
from openpvsg_amd import unitrack as T

torch.manual_seed(0)
sizes = [300, 300, 280, 300, 150, 90, 300, 300, 40, 300, 300, 210, 300, 300, 300, 102]
fn = [torch.nn.functional.normalize(torch.relu(torch.randn(n, 1024, device='cuda')), dim=1) for n in sizes]
print('total', timeit(lambda: T.reconsdot_cost(fn, fn), 10, 3))
Ft = torch.nn.utils.rnn.pad_sequence(fn, batch_first=True)
Nt, Pt, d = Ft.shape
print('pad', timeit(lambda: torch.nn.utils.rnn.pad_sequence(fn, batch_first=True), 10, 3))
A = Ft.reshape(Nt * Pt, d) @ Ft.reshape(Nt * Pt, d).t()
print('gemm A', timeit(lambda: Ft.reshape(Nt * Pt, d) @ Ft.reshape(Nt * Pt, d).t(), 10, 3), A.shape)
S = A * 100
print('softmax rows', timeit(lambda: torch.softmax(S, 1), 10, 3), 'cols', timeit(lambda: torch.softmax(S, 0), 10, 3))
P = torch.softmax(S, 1)
print('num', timeit(lambda: (P * A).view(Nt, Pt, Nt, Pt).sum((1, 3)), 10, 3))
G = Ft @ Ft.transpose(1, 2)
print('gram', timeit(lambda: Ft @ Ft.transpose(1, 2), 10, 3))
P3 = P.view(Nt * Pt, Nt, Pt).transpose(0, 1)
print('bmm td', timeit(lambda: torch.bmm(P3, G), 10, 3))
Q = torch.bmm(P3, G)
print('q reduce', timeit(lambda: (Q * P3).sum(-1).view(Nt, Nt, Pt).sum(-1), 10, 3))
Pc3 = P.view(Nt, Pt, Nt * Pt).transpose(1, 2)
print('bmm dt', timeit(lambda: torch.bmm(Pc3, G), 10, 3))
