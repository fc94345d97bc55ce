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
%
%   The parameter defaulting and the rand() draws below (down to the call of sparse_nmf_mex) repeat, in the reference's
%   order so that MATLAB's generator yields the same initial factors, lines 75-148 of src/sparse_nmf.m of
%   lordet01/SE_SNMF_NAT, which is
%     Copyright (C) 2015 Mitsubishi Electric Research Labs (Jonathan Le Roux, Felix Weninger, John R. Hershey)
%     Licensed under the Apache License, Version 2.0 (http://www.apache.org/licenses/LICENSE-2.0)
%   (J. Le Roux, J. R. Hershey, F. Weninger, "Sparse NMF - half-baked or well done?", MERL TR2015-023, March 2015).
%   This file is a modified version of that block; the solver behind sparse_nmf_mex is new code.
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
if ~isfield(p, 'display'), p.display = 0; end
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
if p.display ~= 0
    % what src/sparse_nmf.m:181-183,266-270,276-278,288-290 print, from the objective vectors the engine recorded (the
    % whole solve is one MEX call, so the lines appear after it instead of during it)
    stopped = n_iter < p.max_iter;
    if gpu_variant   % src/sparse_nmf_GPU.m:266-268,274
        for it = 1:n_iter
            fprintf('iteration %d div = %.3e cost = %.3e\n', it, div(it), cost(it));
        end
        if stopped && p.conv_eps > 0, disp('Convergence reached, aborting iteration'); end
    else
        fprintf(1, 'Performing sparse NMF with beta-divergence, beta=%.1f\n', p.beta);
        if opts.cost_check
            str = [];
            for it = 1:n_iter
                fprintf(repmat('\b', 1, length(str)));
                str = sprintf('iteration %d div = %.3e cost = %.3e', it, div(it), cost(it));
                fprintf('%s', str);
            end
            if stopped && p.conv_eps > 0, disp('Convergence reached, aborting iteration'); end
        end
        disp('\nMax Iteration reached, aborting iteration\n');
    end
end
end
