% godual_ranging_OP_vitesse_hip.m — experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m on the MI355X library: the same script, the
% per-window arithmetic (mean removal, carrier from fft(d1.^2), mix, the velocity-compensated resampling interp1(...) with the carried t0,
% ifft(fft(yi).*fcode), the peak) behind twstft_processing_mex.  Channel 1 (returned signal) is resampled, channel 2 (reference) is not,
% as in the reference (:40-47).  The carried t0 / dt live in the library (twx_set_resample); every record brings its window's dt back.
pkg load signal

fs=5e6;
vitesse=-3.25e-9;                                            % godual_ranging_OP_vitesse.m:4

f=fopen('OP_prn22bpskcode0.bin');
codeb=fread(f,inf,'int8');                                   % :7 — 0/1 bytes; repelems, code-mean(code), conj(fft) happen in the library
fclose(f);
N=2*length(codeb);
freq=linspace(-fs/2,fs/2,N);
k=find((freq<106200)&(freq>96200));                          % :32

twstft_processing_mex('option','replica','unipolar_zero_mean');   % :8-10
filelist=dir('./OP11h45.bin');
for filenum=1:length(filelist)
  name=filelist(filenum).name
  % channel 2, the reference: plain correlation, Nint = 0 (:47)
  twstft_processing_mex('option','vitesse',[0 0 0]);
  [indice2,correction22,~,~,~,~,~,~,xval2]=twstft_processing_mex('file',name,2,2,0,codeb,fs,0);     % df = 0: d2 is not mixed (:47)
  % channel 1, the returned signal: mixed, resampled with the carried offset (:31-43), correlated (:46)
  twstft_processing_mex('option','vitesse',[vitesse 0 0]);
  [indice1,correction12,~,~,df,~,~,~,xval1]=twstft_processing_mex('file',name,2,1,[k(1) k(end)],codeb,fs,0);
  [~,~,~,status,dt]=twstft_processing_mex('extra');
  indice1=indice1+dt;                                        % :68
  printf("%f %f\n",[indice1;indice2]);                       % :50
  solution12=indice1+correction12;                           % :76  (the 3-point polyfit vertex of :56-57 is the closed form the library returns)
  solution22=indice2+correction22;
  subplot(211)
  plot((solution12-solution22)/fs);                          % ranging solution
  xlabel('time (s)'); ylabel('ranging delay (s)')
  [a,b]=polyfit([1:length(solution12)],(solution12-solution22)/fs,1);
  subplot(212)
  plot((solution12-solution22)/fs-b.yf);
  std((solution12-solution22)/fs-b.yf)
  mean((solution12-solution22)/fs-b.yf)
  xlabel('time (s)'); ylabel('delay - parabolic fit (s)')
end
