function [w, h, objective] = sparse_nmf_GPU(v, p)
% SPARSE_NMF_GPU  Drop-in replacement of src/sparse_nmf_GPU.m (same deltas w.r.t. sparse_nmf.m:
% V is not floored, the objective vectors stay zero, cost_check is ignored).  Selected by
% p.useGPU at src/bnmf_sep_event_RT_IS16.m:150-154.
if ~exist('p', 'var'), p = struct; end
p.snmf_gpu_variant = 1;
[w, h, objective] = sparse_nmf(v, p);
end
