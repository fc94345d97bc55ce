function [w, h, objective] = sparse_nmf(v, p)
% SPARSE_NMF  Drop-in replacement of src/sparse_nmf.m running on an MI355X (libsnmf_hip.so).
%
%   Same signature, fields, defaults and error behaviour as the reference function.  This wrapper
%   does what cannot (or should not) cross the C boundary: the defaulting of the parameter struct
%   and the random initial factors, drawn with MATLAB's own generator exactly like the reference
%   (rand('seed', .) then rand(m,r), rand(r,n)), so that RNG parity holds by construction.  The hot
%   loop (normalisation, H / W multiplicative updates, objective, early stop) runs in the MEX
%   function sparse_nmf_mex (integration/sparse_nmf_mex.cpp).
%
%   To emulate src/sparse_nmf_GPU.m set p.snmf_gpu_variant = 1 (no V floor, objective left zero).
m = size(v, 1);
n = size(v, 2);
if ~exist('p', 'var'), p = struct; end
if ~isfield(p, 'max_iter'), p.max_iter = 100; end
if ~isfield(p, 'random_seed'), p.random_seed = 1; end
if ~isfield(p, 'sparsity'), p.sparsity = 0; end
if ~isfield(p, 'conv_eps'), p.conv_eps = 0; end
if ~isfield(p, 'cf'), p.cf = 'kl'; end
switch p.cf
    case 'is', p.beta = 0;
    case 'kl', p.beta = 1;
    case 'ed', p.beta = 2;
    otherwise
        if ~isfield(p, 'beta'), p.beta = 1; end
end
if p.random_seed > 0, rand('seed', p.random_seed); end %#ok<RAND>
if ~isfield(p, 'init_w')
    if ~isfield(p, 'r'), error('Number of components or initialization must be given'); end
    r = p.r;
    w = rand(m, r);
else
    ri = size(p.init_w, 2);
    w(:, 1:ri) = p.init_w;
    if isfield(p, 'r') && ri < p.r
        w(:, (ri + 1):p.r) = rand(m, p.r - ri);
        r = p.r;
    else
        r = ri;
    end
end
if ~isfield(p, 'init_h')
    h = rand(r, n);
elseif ischar(p.init_h) && strcmp(p.init_h, 'ones')
    fprintf('sup_nmf: Initalizing H with ones.\n');
    h = ones(r, n);
else
    h = p.init_h;
end
if ~isfield(p, 'w_update_ind'), p.w_update_ind = true(r, 1); end
if ~isfield(p, 'h_update_ind'), p.h_update_ind = true(r, 1); end
gpu_variant = isfield(p, 'snmf_gpu_variant') && p.snmf_gpu_variant;
opts = struct('beta', p.beta, 'max_iter', p.max_iter, 'conv_eps', p.conv_eps, ...
              'w_update_ind', logical(p.w_update_ind(:)), 'h_update_ind', logical(p.h_update_ind(:)), ...
              'floor_v', double(~gpu_variant), 'device', 0);
if isfield(p, 'snmf_devices') && ~isempty(p.snmf_devices)
    % p.snmf_devices = 0:7 -- the same call with the frames sharded over these GPUs (one MATLAB process, the C ABI's
    % snmf_sparse_nmf_multi_f64): how run_basis_DNMF.m:40,47,53 / run_basis_train.m:88 reach the 8-GPU path unchanged
    opts.devices = double(p.snmf_devices(:)');
end
if gpu_variant
    opts.cost_check = 1;
else
    opts.cost_check = double(p.cost_check ~= 0);   % errors like the reference if the field is absent
end
[w, h, div, cost, n_iter] = sparse_nmf_mex(double(v), double(w), double(h), double(p.sparsity), opts);
objective = struct;
if gpu_variant || ~opts.cost_check
    objective.div = zeros(1, p.max_iter);
    objective.cost = zeros(1, p.max_iter);
elseif n_iter < p.max_iter
    objective.div = div(1:n_iter);
    objective.cost = cost(1:n_iter);
else
    objective.div = div(1:p.max_iter);
    objective.cost = cost(1:p.max_iter);
end
end
