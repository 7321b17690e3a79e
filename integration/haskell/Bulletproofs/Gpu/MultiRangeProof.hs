{-# LANGUAGE RecordWildCards, ScopedTypeVariables #-}
-- | Drop-in for Bulletproofs.MultiRangeProof (Bulletproofs/MultiRangeProof.hs:1-8): `generateProof`,
-- `generateProofUnsafe`, `verifyProof` with the reference's types (and the reference's totality: see `verifyProof`),
-- plus `verifyProofs` = `map verifyProof` as one batched call.  The MonadRandom draws stay in Haskell, in the reference's order, and only happen after the
-- validation of `generateProof` has passed, exactly as in MultiRangeProof/Prover.hs:25-43.
-- NOT COMPILED here (no GHC in the build image).
module Bulletproofs.Gpu.MultiRangeProof
  ( generateProof
  , generateProofUnsafe
  , verifyProof
  , verifyProofs
  ) where

import Protolude
import qualified Data.ByteString as BS
import qualified Data.ByteString.Internal as BI
import Control.Monad.Random (MonadRandom, getRandomR)
import Foreign (withForeignPtr)
import System.IO.Unsafe (unsafePerformIO)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr, _r)
import Data.Curve.Weierstrass (Point(..))

import Bulletproofs.RangeProof.Internal
import Bulletproofs.InnerProductProof.Internal (InnerProductProof(..))
import Bulletproofs.Utils (logBase2, logBase2M, isLogBase2, log2Ceil, chooseBlindingVectors)
import Bulletproofs.Gpu.FFI

-- | MultiRangeProof/Prover.hs:19-43, control flow unchanged: the three guards, then the padding with (0, 0).
generateProof
  :: MonadRandom m
  => Integer
  -> [(Fr, Fr)]
  -> ExceptT (RangeProofError Fr) m (RangeProof Fr PA)
generateProof upperBound vsAndvBlindings = do
  unless (upperBound < fromIntegral _r) $ throwE $ UpperBoundTooLarge upperBound
  case doubleLogM of
     Nothing -> throwE $ NNotPowerOf2 upperBound
     Just n -> do
       unless (checkRanges n vs) $ throwE $ ValuesNotInRange vs
       lift $ generateProofUnsafe upperBound vsAndvBlindingsExp2
  where
    doubleLogM :: Maybe Integer
    doubleLogM = do
      x <- logBase2M upperBound
      logBase2M x >> pure x
    vs = fst <$> vsAndvBlindings
    m = length vsAndvBlindings
    residue = replicate (2 ^ log2Ceil m - m) (0, 0)
    vsAndvBlindingsExp2 = vsAndvBlindings ++ residue

-- | MultiRangeProof/Prover.hs:47-123.  Draw order of the reference: chooseBlindingVectors nm (:65), aBlinding,
-- sBlinding (:69-70), t1Blinding, t2Blinding (:81-82); the draws do not depend on anything computed in between, so
-- they are all taken first and handed to the engine.
generateProofUnsafe
  :: forall m . MonadRandom m
  => Integer
  -> [(Fr, Fr)]
  -> m (RangeProof Fr PA)
generateProofUnsafe upperBound vsAndvBlindings = do
  let n  = logBase2 upperBound
      m  = fromIntegral (length vsAndvBlindings) :: Integer
      nm = n * m
      genBlinding = (fromInteger :: Integer -> Fr) <$> getRandomR (1, fromIntegral _r - 1)
  (sL :: [Fr], sR :: [Fr]) <- chooseBlindingVectors nm
  aBlinding  <- genBlinding
  sBlinding  <- genBlinding
  t1Blinding <- genBlinding
  t2Blinding <- genBlinding
  let k = log2Ceil (fromIntegral nm)
  pure $ unsafePerformIO $ do
    sc  <- BI.mallocByteString (5 * 32)
    pts <- BI.mallocByteString ((4 + 2 * k) * 64)
    withForeignPtr sc $ \psc -> withForeignPtr pts $ \ppt ->
      withBS (encFrs (fst <$> vsAndvBlindings)) $ \pv -> withBS (encFrs (snd <$> vsAndvBlindings)) $ \pg ->
      withBS (encFrs sL) $ \psl -> withBS (encFrs sR) $ \psr ->
      withBS (encFr aBlinding) $ \pa -> withBS (encFr sBlinding) $ \pr ->
      withBS (encFr t1Blinding) $ \p1 -> withBS (encFr t2Blinding) $ \p2 ->
        withCtx (\cx -> c_rangeproof_prove cx (fromIntegral n) (fromIntegral m) pv pg psl psr pa pr p1 p2 0 psc ppt
                          >>= check cx "bp_rangeproof_prove")
    let scalars = BI.fromForeignPtr sc 0 (5 * 32)
        points  = BI.fromForeignPtr pts 0 ((4 + 2 * k) * 64)
        sAt i = decFr (BS.take 32 (BS.drop (32 * i) scalars))
        pAt i = decPt (BS.take 64 (BS.drop (64 * i) points))
    pure RangeProof
      { tBlinding = sAt 0, mu = sAt 1, t = sAt 2
      , aCommit = pAt 0, sCommit = pAt 1, t1Commit = pAt 2, t2Commit = pAt 3
      , productProof = InnerProductProof
          { lCommits = pAt <$> [4 .. 3 + k], rCommits = pAt <$> [4 + k .. 3 + 2 * k], l = sAt 3, r = sAt 4 } }

-- | MultiRangeProof/Verifier.hs:19-39, total like the reference's function: any `upperBound` (n = logBase2 upperBound is
-- whatever Double arithmetic gives, Utils.hs:93-94), any number of commitments, any number of L/R pairs in the proof.
-- bp_rangeproof_verify routes honest-shape proofs to the batch engine and follows the reference's definitions
-- literally for everything else (prop_differentUpperBound, test/Test/Protocol/RangeProof.hs:173-181).  Where the
-- reference's evaluation raises (`challenges !! j` with too few pairs once verifyTPoly holds, `2 ^ (-1)` for no
-- commitments) the library answers BP_ERR_REFERENCE_FAILS and this raises too; no Bool is invented.
verifyProof :: Integer -> [PA] -> RangeProof Fr PA -> Bool
verifyProof upperBound vCommits RangeProof{..} = unsafePerformIO $ do
  let InnerProductProof{..} = productProof
      k       = min (length lCommits) (length rCommits)     -- zip lCommits rCommits (InnerProductProof/Verifier.hs:53)
      ub      = leBytes (max 1 (byteLength upperBound)) (max 0 upperBound)
      scalars = encFrs [tBlinding, mu, t, l, r]
      points  = encPts ([aCommit, sCommit, t1Commit, t2Commit] <> take k lCommits <> take k rCommits)
  when (upperBound < 0) $ panic "verifyProof: negative upper bound (logBase2 of a negative number is NaN)"
  out <- BI.create 1 $ \po -> withBS ub $ \pu -> withBS (encPts vCommits) $ \pv ->
    withBS scalars $ \ps -> withBS points $ \pp ->
      withCtx (\cx -> c_rangeproof_verify cx pu (fromIntegral (BS.length ub)) pv (fromIntegral (length vCommits)) ps pp
                                          (fromIntegral k) po >>= check cx "bp_rangeproof_verify")
  pure (BS.head out /= 0)
  where
    byteLength :: Integer -> Int
    byteLength v = length (takeWhile (> 0) (iterate (`div` 256) v))

-- | `map (\(vs, p) -> verifyProof upperBound vs p)` with the honest-shape proofs of the list verified as ONE batched
-- call.  The batch engine takes one shape per call: n = logBase2 upperBound a power of two <= 128, the commitments
-- padded to m2 = 2 ^ log2Ceil (length vs) of the first proof, and exactly k = log2Ceil (n * m2) L/R pairs (what
-- generateProofUnsafe produces).  A proof of any other shape - another padded commitment count, lCommits or rCommits
-- of another length, or any proof at all when n is outside the engine's domain - is never sent with the batch (the
-- library would read its rows at the wrong offsets): it goes through `verifyProof` above, one call each, which is
-- total.  The verdicts are per proof (no random linear combination across proofs) and come back in list order.
verifyProofs :: Integer -> [([PA], RangeProof Fr PA)] -> [Bool]
verifyProofs _ [] = []
verifyProofs upperBound batch@((vs0, _) : _) = unsafePerformIO $ do
  let n       = logBase2 upperBound                          -- Verifier.hs:34
      nOk     = upperBound > 0 && n >= 1 && n <= 128 && isLogBase2 n
      m       = if null vs0 then 0 else 2 ^ log2Ceil (length vs0) :: Int   -- padding with mempty, Verifier.hs:37-38
      k       = log2Ceil (fromIntegral n * m)                -- as generateProofUnsafe, MultiRangeProof/Prover.hs
      ipp     = productProof
      fits (vs, p) = nOk && m >= 1 && not (null vs) && 2 ^ log2Ceil (length vs) == m
                     && length (lCommits (ipp p)) == k && length (rCommits (ipp p)) == k
      tagged  = [ (fits e, e) | e <- batch ]
      inBatch = [ e | (True, e) <- tagged ]
      b       = length inBatch
      proofs  = snd <$> inBatch
      row f   = BS.concat (f <$> proofs)
      at xs i = fromMaybe O (atMay xs i)
      scalars = BS.concat [ row (encFr . tBlinding), row (encFr . mu), row (encFr . t)
                          , row (encFr . l . ipp), row (encFr . r . ipp) ]
      points  = BS.concat $
                  [ row (encPt . aCommit), row (encPt . sCommit), row (encPt . t1Commit), row (encPt . t2Commit) ]
                  <> [ row (\p -> encPt (lCommits (ipp p) `at` i)) | i <- [0 .. k - 1] ]
                  <> [ row (\p -> encPt (rCommits (ipp p) `at` i)) | i <- [0 .. k - 1] ]
                  <> [ BS.concat [ encPt (vs `at` j) | (vs, _) <- inBatch ] | j <- [0 .. m - 1] ]
  batched <- if b == 0 then pure [] else do
    out <- BI.create b $ \po -> withBS scalars $ \ps -> withBS points $ \pp ->
      withCtx (\cx -> c_rangeproof_verify_batch cx (fromIntegral n) (fromIntegral m) (fromIntegral b) ps pp po
                        >>= check cx "bp_rangeproof_verify_batch")
    pure ((/= 0) <$> BS.unpack out)
  -- merge back in list order; the out-of-shape proofs are evaluated lazily, one total call each
  let merge [] _ = []
      merge ((True, _) : rest) (v : vs) = v : merge rest vs
      merge ((True, _) : _) [] = panic "verifyProofs: verdict count"
      merge ((False, (vs, p)) : rest) vsB = verifyProof upperBound vs p : merge rest vsB
  pure (merge tagged batched)
