function NTF_sep_event_RT(path_in, path_event, path_noise, path_denoise, B_Mel_x, B_Mel_d, B_DFT_x, B_DFT_d, p) %#ok<INUSL>
% NTF_sep_event_RT  Drop-in for src/NTF_sep_event_RT.m (p.NMF_algorithm = 'SNMF', one channel) that runs the
% whole frame loop -- init_buff + bnmf_sep_event_RT_IS16 per hop + overlap-add -- on the GPU through
% snmf_online_mex (integration/snmf_online_mex.cpp, C ABI include/snmf.h: snmf_online_*).
% Same signature, same files written: the denoised PCM (fwrite int16, then pcm2wav) and B_D_u.mat.
% Put this directory first on the MATLAB path.

if p.adapt_train_N && exist('B_D_u.mat', 'file')      % src/NTF_sep_event_RT.m:28-38
    try
        S = load('B_D_u.mat');
        B_DFT_d = S.B_DFT_d;
    catch
    end
end

fin = fopen(path_in(1,:), 'rb');
fread(fin, 22, 'int16');                               % skip the wav header (:56-59)
pcm = fread(fin, inf, 'int16');
fclose(fin);

% The reference's random draws, made here with MATLAB's own generator and IN THE REFERENCE'S ORDER, so that they are
% the reference's values:
%   1. init_buff (src/NTF_sep_event_RT.m:60 -> src/init_buff.m:38-39) draws g.A_d = rand(R_d, m) and then
%      g.Ad_blk = rand(p.R_a, p.m_a) from the global stream AS THE CALLER LEFT IT -- nothing has re-seeded it yet;
%   2. only the first sparse_nmf call of the frame loop seeds the legacy generator (src/sparse_nmf.m:112-114) and
%      draws H0 = rand(r, 1) (:133-134) -- the same vector every frame, because every call re-seeds.
% g.A_d is never read before it is overwritten (src/bnmf_sep_event_RT_IS16.m:22, :395); it is drawn only to leave
% the stream where the reference leaves it for Ad_blk.
A_d0 = rand(size(B_DFT_d,2), p.blk_len_sep);           %#ok<NASGU>  src/init_buff.m:38
Ad_blk0 = rand(p.R_a, p.m_a);                          % src/init_buff.m:39
if p.random_seed > 0
    rand('seed', p.random_seed);                       %#ok<RAND>  src/sparse_nmf.m:112-114
end
H0 = rand(size(B_DFT_x,2) + size(B_DFT_d,2), 1);      % src/sparse_nmf.m:133-134

h = snmf_online_mex('create', B_DFT_x, B_DFT_d, H0, Ad_blk0, p);
x = snmf_online_mex('process', h, pcm, 1);             % every hop + the delay+1 end-of-file frames (:67-76)
B_DFT_d = snmf_online_mex('basis', h, size(B_DFT_d,1), size(B_DFT_d,2)); %#ok<NASGU>
snmf_online_mex('destroy', h);

fout = fopen(path_denoise(1,:), 'wb');
fwrite(fout, x, 'int16');                              % :124
fclose(fout);
B_Mel_d = B_DFT_d;                                     %#ok<NASGU>  DFT mode: the Mel slots hold the DFT bases
save('B_D_u.mat', 'B_DFT_d', 'B_Mel_d');               % :138-140
pcm2wav(path_denoise(1,:), p);                         % :159
end
