% processing_hip.m — drop-in for processing(d,k) of processing/Octave/godual_ranging.m:12 that
% takes the raw int16 window instead of the mean-removed complex vector (mean removal,
% godual_ranging.m:80, happens on the GPU).  Globals as in the reference script (:3).
%   raw   : int16 column as read by fread(f, length(fcode)*2*nchan, 'int16=>int16')
%   chan  : 1-based channel, or 0 = every channel from one upload (outputs nchan x nwin)
%   k     : search band indices as produced by find((freq<20000)&(freq>-20000)) (:83)
%   codeb : code file bytes (before repelems, :63)
function [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval]=processing_hip(raw,nchan,chan,k,codeb)
  global fs Nint
  [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval] = ...
      twstft_processing_mex(raw, nchan, chan, [k(1) k(end)], codeb, fs, Nint);
end
